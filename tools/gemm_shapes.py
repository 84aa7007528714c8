#!/usr/bin/env python3
"""Which matrix products a workload of tools/workloads.py spends its time in: every rlgan_dense_* call of one eager
iteration timed with events, grouped by (entry point, m, k, n).  A development aid."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import rl_gan_amd as R
from rl_gan_amd import _lib
import workloads

name = sys.argv[1] if len(sys.argv) > 1 else "pixel_curious_gan_universe_42_b4096"
records = []


def wrap(fname, dims):
    real = getattr(_lib.lib, fname)

    def f(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = real(*a)
        e1.record()
        records.append((fname, tuple(a[i] for i in dims), e0, e1))
        return rc
    setattr(_lib.lib, fname, f)


entry = [w for w in workloads.WORKLOADS if w[0] == name][0]
build, batch = entry[1], entry[2] or 4096
if name.startswith("pixel"):
    # the graph would hide the calls: build the trainer by hand the way the workload does, without capture
    import types
    orig = R.PixelGanTrainer.capture
    R.PixelGanTrainer.capture = lambda self: None
step = build(batch)
for _ in range(3):
    step()
torch.cuda.synchronize()
wrap("rlgan_dense_forward", (5, 6, 7))            # m, k, n
wrap("rlgan_dense_backward_input", (5, 6, 7))
wrap("rlgan_dense_backward_input_split", (5, 6, 7))
wrap("rlgan_dense_backward_weights", (5, 6, 7))
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for fname, dims, e0, e1 in records:
    t = e0.elapsed_time(e1) * 1e3
    k = (fname.replace("rlgan_dense_", ""), dims)
    c, s = agg.get(k, (0, 0.0))
    agg[k] = (c + 1, s + t)
tot = sum(s for _, s in agg.values())
for (fn, (m, k, n)), (c, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * m * k * n * c
    print(f"{fn:18s} m={m:6d} k={k:6d} n={n:5d}  calls {c:3d}  total {s:8.1f} us  ({fl / s / 1e6:6.1f} TFLOP/s)")
print(f"all products: {tot:.1f} us in {len(records)} calls")
