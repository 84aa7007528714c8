// Host-side glue shared by the translation units of librlgan_amd.so.
#pragma once
#include "../../include/rlgan_amd.h"
