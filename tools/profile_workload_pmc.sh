#!/bin/bash
# Matrix-pipe counters per kernel (template instance) of one workload of tools/workloads.py, next to its kernel trace
# (tools/profile_workloads.sh must have run for the same tag: its kernel_stats.csv supplies the durations).
#   bash tools/profile_workload_pmc.sh r03f_pixel_b4096 pixel_curious_gan_universe_42_b4096
set -euo pipefail
TAG=${1:?tag}; WL=${2:?workload name}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
test -f "$OUT/kernel_stats.csv"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/workloads.py 4096 $WL"
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d "$OUT/mfma" -o mfma -- $CMD > "$OUT/mfma.log" 2>&1
cd "$ROOT"
python3 tools/mfma_json.py --full-names "$(ls $OUT/mfma/*.db | head -1)" "$OUT/kernel_stats.csv" "python3 tools/workloads.py 4096 $WL" > "$OUT/pmc_mfma.json"
rm -rf "$OUT/mfma"
python3 - "$OUT/pmc_mfma.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, e in sorted(d.items(), key=lambda kv: -kv[1].get("avg_duration_us", 0))[:25]:
    print(f"{k[:70]:70s} {e.get('avg_duration_us', 0):8.1f} us  busy {e.get('mfma_busy_frac', 0):.3f}  valu/mfma {e.get('valu_per_mfma', 0)}")
PY
