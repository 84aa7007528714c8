#!/bin/bash
# Kernel-trace summary of the bench command (first lines), run on the GPU box from the repository root:
#   bash tools/trace_bench.sh <tag> [bench args]
set -euo pipefail
TAG=${1:?tag}; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/trace.log"
cd "$ROOT"
python3 tools/rocpd_stats.py "$(ls $OUT/trace/*.db | head -1)" > "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
cut -c1-60 "$OUT/bench.json"; sed -n 1,7p "$OUT/kernel_stats.csv" | sed 's/(anonymous namespace):://g' | cut -c1-150
