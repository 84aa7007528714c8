#!/usr/bin/env python3
"""Matrix-pipe counters per launch and kernel from one rocprofv3 counter run (rocpd sqlite output) and the kernel-trace
summary of the same command:

    python tools/mfma_json.py pmc_results.db kernel_stats.csv "<command that was profiled>" > profiles/rNN_vM_pmc_mfma.json

Counter pass: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE (eight SQ counters and one GRBM counter: one pass).

Derived per kernel:
  mfma_busy_frac   SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x average duration from the trace x 2.4 GHz): the share of the
                   chip's matrix-pipe cycles AT PEAK CLOCK that held an MFMA -- the figure to set beside a roofline
                   fraction priced against the 2.4 GHz peak (the chip may clock lower under load, so this is a floor of
                   the busy share of actual cycles)
  mfma_flop        SQ_INSTS_VALU_MFMA_MOPS_F32 x 512: floating-point operations the matrix pipe executed (padding included)
  valu_per_mfma    SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU counts the MFMAs too)"""
import csv
import json
import re
import sqlite3
import sys

PEAK_HZ, SIMDS = 2.4e9, 1024
FULL = "--full-names" in sys.argv          # key kernels by name AND template arguments (a workload with many instances of one template)
if FULL:
    sys.argv.remove("--full-names")


def key_of(name):
    if not FULL:
        m = re.search(r"(\w+_kernel)", name)
        return m.group(1) if m else None
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").strip()
    n = n.split("(")[0].strip()
    return n if n.endswith(">") or n.endswith("_kernel") else None


db = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
kn = "kernel_name" if "kernel_name" in cols else "name"
per = {}
for name, counter, total, n in db.execute(f"select {kn}, counter_name, sum(value), count(*) from counters_collection group by {kn}, counter_name"):
    k = key_of(name)
    if k:
        per.setdefault(k, {})[counter] = total / n
dur = {}
with open(sys.argv[2]) as f:
    for row in csv.DictReader(f):
        k = key_of(row["Name"])
        if k:
            dur[k] = float(row["AverageUs"]) * 1e3
kernels = {}
for k, c in sorted(per.items()):
    e = {"counters_per_launch": {n: round(v, 1) for n, v in sorted(c.items())}}
    if k in dur and dur[k] > 0:
        e["avg_duration_us"] = round(dur[k] / 1e3, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * dur[k] * 1e-9 * PEAK_HZ), 4)
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in c:
        e["mfma_flop"] = int(c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512)
    if c.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"], 3)
    kernels[k] = e
print(json.dumps({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA "
                             "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- " + sys.argv[3],
                  "note": "per-launch averages over all dispatches of the run; durations from the kernel-trace pass of the same command "
                          "(counter passes run slower and are not timed); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                          "duration x 2.4 GHz)",
                  "kernels": kernels}, indent=1))
