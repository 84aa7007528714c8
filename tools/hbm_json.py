#!/usr/bin/env python3
"""HBM bytes per launch and kernel from two rocprofv3 counter runs (rocpd sqlite output), one with
`--pmc FETCH_SIZE`, one with `--pmc WRITE_SIZE`:

    python tools/hbm_json.py fetch_results.db write_results.db "<command that was profiled>" > profiles/rNN_pmc_hbm.json

Both counters are in KB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled;
WRITE_SIZE is taken as is."""
import json
import re
import sqlite3
import sys


COUNTS = {}


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    for name, total, n in db.execute(f"select {kn}, sum(value), count(*) from counters_collection where counter_name = ? group by {kn}",
                                     (counter,)):
        m = re.search(r"(\w+_kernel)", name)
        if m:
            out[m.group(1)] = total / n
            COUNTS[m.group(1)] = n
    return out


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) & set(write)):
    kernels[k] = {"fetch_kb_per_launch": round(fetch[k], 1), "write_kb_per_launch": round(write[k], 1),
                  "hbm_bytes_per_launch": int(round((2.0 * fetch[k] + write[k]) * 1024))}
# one training iteration = the launches between two phase-1 row passes: every launch of the step kernels divided by the
# number of iterations; the tail launch (rollout_eval_kernel) at its per-launch average, which the acting-only launches
# that fill the ring pull down a little
n_it = COUNTS.get("phase1_kernel", 0)
iteration = None
if n_it:
    iteration = sum(kernels[k]["hbm_bytes_per_launch"] * COUNTS[k] / n_it for k in ("phase1_kernel", "phase2_kernel", "wpass_kernel", "apply_kernel")
                    if k in kernels)
    iteration = int(iteration + kernels.get("rollout_eval_kernel", {}).get("hbm_bytes_per_launch", 0))
print(json.dumps({"iteration_hbm_bytes": iteration, "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + sys.argv[3],
                  "note": "per-launch averages over all dispatches of the run (for rollout_eval_kernel that includes the acting-only "
                          "launches that fill the ring); both counters are in KB; FETCH_SIZE is doubled (on gfx950 it tallies "
                          "128-byte requests at 64 bytes), WRITE_SIZE taken as is",
                  "kernels": kernels}, indent=1))
