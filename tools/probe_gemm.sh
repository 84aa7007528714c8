#!/bin/bash
# tools/gemm_slabs.py on a list of shapes with each of the given builds of the library in turn:
#   bash tools/probe_gemm.sh "gpurun_ab/lib_a.so gpurun_ab/lib_b.so" "4096 2592 256" ...
set -euo pipefail
LIBS=$1; shift
cp rl-gan_amd/librlgan_amd.so /tmp/lib_a.so
for l in $LIBS; do
  cp "$l" rl-gan_amd/librlgan_amd.so
  for s in "$@"; do
    echo "== $l $s"
    timeout -k 10 120 python3 tools/gemm_slabs.py $s 2>&1 | grep "suggested" | cut -c1-150
  done
done
cp /tmp/lib_a.so rl-gan_amd/librlgan_amd.so
