#!/bin/bash
# Builds librlgan_amd.so (gfx950) in-tree, next to the sources.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 ${RLGAN_EXTRA_FLAGS:-}"   # RLGAN_EXTRA_FLAGS: experiments only (e.g. -DRLGAN_GD=3)
mkdir -p build
pids=()
for f in learn_kernels rollout_kernels replay_kernels host_api conv_kernels dense_kernels deconv_kernels deconv_fused_kernels norm_kernels pixel_mlp_kernels; do
  rm -f build/$f.o
  $HIPCC $FLAGS -c $f.hip -o build/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done      # a bare `wait` would hide a failed compile
$HIPCC --offload-arch=gfx950 -shared -fPIC build/learn_kernels.o build/rollout_kernels.o build/replay_kernels.o build/host_api.o build/conv_kernels.o build/dense_kernels.o build/deconv_kernels.o build/deconv_fused_kernels.o build/norm_kernels.o build/pixel_mlp_kernels.o -o ../librlgan_amd.so
echo "built $(cd .. && pwd)/librlgan_amd.so"
