#!/bin/bash
# Kernel-trace summary of the pixel variant's learning iteration (tools/pixelbench.py) for profiles/.
#   bash tools/profile_pixel.sh r01_pixel 4096
set -euo pipefail
TAG=${1:?tag}; B=${2:-4096}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $ROOT/tools/pixelbench.py $B > "$OUT/pixelbench.txt" 2> "$OUT/pixelbench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 $ROOT/tools/pixelbench.py $B > "$OUT/trace.log" 2>&1
cd "$ROOT"
python3 tools/rocpd_stats.py "$(ls $OUT/trace/*.db | head -1)" > "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
cat "$OUT/pixelbench.txt"; head -14 "$OUT/kernel_stats.csv" | cut -c1-160
