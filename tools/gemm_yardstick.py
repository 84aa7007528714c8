#!/usr/bin/env python3
"""A yardstick for the dense products: torch.mm (rocBLAS / hipBLASLt, float32) on the three products of a layer shape,
timed inside a replayed graph like tools/gemm_slabs.py:  python tools/gemm_yardstick.py [m k n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from kbench import timeit_graph

torch.backends.cuda.matmul.allow_tf32 = False
m, k, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 2592, 256)
x = torch.randn(m, k, device="cuda"); w = torch.randn(k, n, device="cuda") * 0.02; dz = torch.randn(m, n, device="cuda")
y = torch.empty(m, n, device="cuda"); dx = torch.empty(m, k, device="cuda"); gw = torch.empty(k, n, device="cuda")
for name, f, fl in (("forward", lambda: torch.mm(x, w, out=y), 2.0 * m * k * n),
                    ("input", lambda: torch.mm(dz, w.t(), out=dx), 2.0 * m * k * n),
                    ("weights", lambda: torch.mm(x.t(), dz, out=gw), 2.0 * m * k * n)):
    f(); torch.cuda.synchronize()
    t = min(timeit_graph(f, reps=10) for _ in range(3))
    print(f"{name:8s} m={m} k={k} n={n}  torch.mm {t:7.1f} us  {fl / t * 1e-6:6.1f} TFLOP/s")
