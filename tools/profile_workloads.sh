#!/bin/bash
# Kernel-trace summary of one of the pixel workloads of tools/workloads.py for profiles/.
#   bash tools/profile_workloads.sh r02_pixel pixel_curious_gan_universe_42_b4096
set -euo pipefail
TAG=${1:?tag}; WL=${2:?workload name}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 $ROOT/tools/workloads.py 4096 $WL > "$OUT/workload.json" 2> "$OUT/trace.log"
cd "$ROOT"
python3 tools/rocpd_stats.py "$(ls $OUT/trace/*.db | head -1)" > "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
cat "$OUT/workload.json"; head -30 "$OUT/kernel_stats.csv" | cut -c1-200
