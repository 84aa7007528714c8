#!/bin/bash
# One round of evidence for profiles/: the bench line, the rocprofv3 kernel-trace summary of the same command, and
# the HBM traffic from two separate counter passes and the matrix-pipe counters from a fourth run.  Run on the GPU box from the repository root:
#   bash tools/profile_round.sh r01_v10
set -euo pipefail
TAG=${1:?tag}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 200 --warmup 20 --no-other-workloads"
timeout -k 10 200 $CMD > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$OUT/mfma" -o mfma -- $CMD > "$OUT/mfma.log" 2>&1
cd "$ROOT"
python3 tools/rocpd_stats.py "$(ls $OUT/trace/*.db | head -1)" > "$OUT/kernel_stats.csv"
python3 tools/hbm_json.py "$(ls $OUT/fetch/*.db | head -1)" "$(ls $OUT/write/*.db | head -1)" "python3 bench.py --steps 200 --warmup 20 --no-other-workloads" > "$OUT/pmc_hbm.json"
python3 tools/mfma_json.py "$(ls $OUT/mfma/*.db | head -1)" "$OUT/kernel_stats.csv" "python3 bench.py --steps 200 --warmup 20 --no-other-workloads" > "$OUT/pmc_mfma.json"
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write" "$OUT/mfma"
tail -c 600 "$OUT/bench.json"; head -8 "$OUT/kernel_stats.csv"
