#!/bin/bash
# Like tools/ab_bench.sh for one of the workloads of tools/workloads.py: A B A B on one box.
#   bash tools/ab_workload.sh gpurun_ab/lib_old.so gpurun_ab/lib_new.so pixel_curious_gan_universe_42_b4096
set -e
A=${1:?first library}; B=${2:?second library}; W=${3:-pixel_curious_gan_universe_42_b4096}
for i in 1 2; do
  for v in "$A" "$B"; do
    cp "$v" rl-gan_amd/librlgan_amd.so
    echo "$v $(timeout -k 10 200 python tools/workloads.py 4096 $W 2>/dev/null | grep ms_per_step)"
  done
done
