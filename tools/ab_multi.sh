#!/bin/bash
# Several builds of the library, one command, one box: each library in turn, the whole round REP times.
#   REP=2 bash tools/ab_multi.sh "python tools/firstlayer.py" gpurun_ab/a.so gpurun_ab/b.so ...
set -e
CMD=${1:?command}; shift
for i in $(seq 1 ${REP:-2}); do
  for v in "$@"; do
    cp "$v" rl-gan_amd/librlgan_amd.so
    echo "== $v"
    timeout -k 10 300 $CMD 2>/dev/null
  done
done
