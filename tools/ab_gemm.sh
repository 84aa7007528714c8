#!/bin/bash
# tools/gemm_slabs.py on a list of shapes with the in-tree library and then with another build of it (A B A):
#   bash tools/ab_gemm.sh gpurun_ab/lib_x.so "4096 2592 256" "4096 512 500"
set -euo pipefail
ALT=$1; shift
cp rl-gan_amd/librlgan_amd.so /tmp/lib_a.so
for pass in A B A; do
  if [ $pass = A ]; then cp /tmp/lib_a.so rl-gan_amd/librlgan_amd.so; else cp "$ALT" rl-gan_amd/librlgan_amd.so; fi
  for s in "$@"; do
    echo "== $pass $s"
    timeout -k 10 120 python3 tools/gemm_slabs.py $s 2>&1 | grep "suggested"
  done
done
cp /tmp/lib_a.so rl-gan_amd/librlgan_amd.so
