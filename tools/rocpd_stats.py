#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output) as CSV.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.csv
"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
for name, calls, total, avg, pct in cur:
    w.writerow([name, calls, round(total, 3), round(avg, 3), round(pct, 3)])
