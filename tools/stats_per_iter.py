#!/usr/bin/env python3
"""Per-iteration view of a kernel_stats.csv of a workload trace: calls, average and microseconds per iteration of the
top kernels (iterations counted from a kernel that runs a known number of times per iteration).
  python tools/stats_per_iter.py <kernel_stats.csv> [kernel-name-substring=apply_kernel] [its calls per iteration=10] [rows=32]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "apply_kernel"
per = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
top = int(sys.argv[4]) if len(sys.argv) > 4 else 32
it = max(int(r["Calls"]) for r in rows if key in r["Name"]) / per
tot = sum(float(r["TotalDurationUs"]) for r in rows)
print(f"iterations {it:.1f}  kernel time per iteration {tot / it:.1f} us  launches per iteration {sum(int(r['Calls']) for r in rows) / it:.1f}")
for r in rows[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    print(f"{name:64s} calls/it {int(r['Calls']) / it:5.1f}  avg {float(r['AverageUs']):7.1f} us  per it {float(r['TotalDurationUs']) / it:7.1f} us")
