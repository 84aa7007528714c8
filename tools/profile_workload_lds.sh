#!/bin/bash
# LDS counters per kernel of one workload of tools/workloads.py (one counter pass of its own).
#   bash tools/profile_workload_lds.sh r03_lds pixel_curious_gan_universe_42_b4096
set -euo pipefail
TAG=${1:?tag}; WL=${2:?workload name}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d "$OUT/lds" -o lds -- python3 $ROOT/tools/workloads.py 4096 $WL > "$OUT/lds.log" 2>&1
cd "$ROOT"
python3 - "$(ls $OUT/lds/*.db | head -1)" <<'PY'
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t.startswith("counters_collection")][0]
cols = [d[0] for d in db.execute(f"select * from {view} limit 1").description]
kn = "kernel_name" if "kernel_name" in cols else "name"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for name, cname, val, did in db.execute(f"select {kn}, counter_name, value, dispatch_id from {view}"):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    agg[n][cname] += val
    if (did, n) not in seen:
        seen.add((did, n)); cnt[n] += 1
rows = []
for n, c in agg.items():
    k = cnt[n]
    busy = c.get("SQ_BUSY_CU_CYCLES", 0) / k
    rows.append((busy, n, k, c))
for busy, n, k, c in sorted(rows, reverse=True)[:24]:
    g = lambda x: c.get(x, 0) / k
    print(f"{n[:64]:64s} n={k:4d} lds_idx_active/cu_busy {g('SQ_LDS_IDX_ACTIVE') / max(1, busy):5.2f}  bank_conflict/idx_active {g('SQ_LDS_BANK_CONFLICT') / max(1, g('SQ_LDS_IDX_ACTIVE')):5.2f}  "
          f"lds insts/mfma {g('SQ_INSTS_LDS') / max(1, g('SQ_INSTS_MFMA')):5.2f}  active_inst_lds/wave_cycles {g('SQ_ACTIVE_INST_LDS') / max(1, g('SQ_WAVE_CYCLES')):5.3f}")
PY
rm -rf "$OUT/lds"
