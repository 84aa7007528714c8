#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 --pmc run (rocpd sqlite output).
usage: pmc_report.py <results.db> <kernel-name-substring>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", tabs)
    sys.exit(1)
cols = [d[0] for d in db.execute(f"select * from {view} limit 1").description]
kn = "kernel_name" if "kernel_name" in cols else "name"
rows = db.execute(f"select {kn}, counter_name, sum(value), count(*) from {view} where {kn} like ? group by {kn}, counter_name",
                  (f"%{pat}%",)).fetchall()
for k, c, v, n in rows:
    print(f"{k[:60]:60s} {c:40s} total {v:16.0f}  per-dispatch {v / n:14.1f}  (n={n})")
