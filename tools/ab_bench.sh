#!/bin/bash
# Two builds of the library against each other on ONE box (clocks differ from box to box by more than most changes
# are worth): alternating runs of the headline bench, A B A B.  Run on the GPU box from the repository root:
#   bash tools/ab_bench.sh gpurun_ab/lib_old.so gpurun_ab/lib_new.so
# The library in place is overwritten; the last run leaves B there.
set -e
A=${1:?first library}; B=${2:?second library}
for i in 1 2; do
  for v in "$A" "$B"; do
    cp "$v" rl-gan_amd/librlgan_amd.so
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-other-workloads --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernel_ms'])"
  done
done
