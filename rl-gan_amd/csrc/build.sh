#!/bin/bash
# Builds librlgan_amd.so (gfx950) in-tree, next to the sources.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
$HIPCC $FLAGS -c learn_kernels.hip -o build/learn_kernels.o &
$HIPCC $FLAGS -c rollout_kernels.hip -o build/rollout_kernels.o &
$HIPCC $FLAGS -c replay_kernels.hip -o build/replay_kernels.o &
$HIPCC $FLAGS -c host_api.hip -o build/host_api.o &
$HIPCC $FLAGS -c conv_kernels.hip -o build/conv_kernels.o &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC build/learn_kernels.o build/rollout_kernels.o build/replay_kernels.o build/host_api.o build/conv_kernels.o -o ../librlgan_amd.so
echo "built $(cd .. && pwd)/librlgan_amd.so"
