#!/bin/bash
# Two builds of the library against each other on ONE box for any command: A B A B.
#   bash tools/ab_run.sh gpurun_ab/lib_old.so gpurun_ab/lib_new.so python tools/convlayers.py
set -e
A=${1:?first library}; B=${2:?second library}; shift 2
for i in 1 2; do
  for v in "$A" "$B"; do
    cp "$v" rl-gan_amd/librlgan_amd.so
    echo "== $v"
    timeout -k 10 300 "$@" 2>/dev/null
  done
done
