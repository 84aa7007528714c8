#!/usr/bin/env python3
"""Per kernel of a `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD` run: launches,
instructions per launch and the share of vector-issue slots that are not MFMAs (an fp32 MFMA holds the slot for 32 cycles,
another vector instruction for about 4).  usage: issue_mix.py <results.db>"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
K = collections.defaultdict(dict)
for k, c, v, n in rows:
    K[k][c] = v
    K[k]["n"] = n
out = []
for k, d in K.items():
    mf, va = d.get("SQ_INSTS_MFMA", 0), d.get("SQ_INSTS_VALU", 0)
    other = max(va - mf, 0)
    slots = 32 * mf + 4 * other
    out.append((slots, k, d))
for slots, k, d in sorted(out, reverse=True)[:30]:
    mf, va = d.get("SQ_INSTS_MFMA", 0), d.get("SQ_INSTS_VALU", 0)
    other = max(va - mf, 0)
    print(f"{k[:90]:90s} n={d['n']:5d} mfma {mf / d['n']:12.0f} other-valu {other / d['n']:12.0f} lds {d.get('SQ_INSTS_LDS', 0) / d['n']:10.0f}"
          f"  non-MFMA share of issue {4 * other / max(slots, 1):5.2f}")
