#!/usr/bin/env python3
"""Time of the three dense products of one layer shape as a function of the number of reduction slabs (the count
rlgan_dense_slabs suggests is marked): python tools/gemm_slabs.py [m k n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from rl_gan_amd import _lib
from kbench import timeit_graph

m, k, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 2592, 256)
lib, st = _lib.lib, torch.cuda.current_stream().cuda_stream
x = torch.randn(m, k, device="cuda"); wb = torch.randn(k + 1, n, device="cuda") * 0.02
dz = torch.randn(m, n, device="cuda"); y = torch.empty(m, n, device="cuda"); dx = torch.empty(m, k, device="cuda"); gw = torch.empty(k + 1, n, device="cuda")
slabs = torch.empty(64 * max(m * n, m * k, (k + 1) * n), device="cuda")


def S():
    return torch.cuda.current_stream().cuda_stream


prods = (("forward", m, n, k + 1, lambda ns: lib.rlgan_dense_forward(x.data_ptr(), k, wb.data_ptr(), y.data_ptr(), n, m, k, n, 1, slabs.data_ptr(), ns, S())),
         ("input", m, k, n, lambda ns: lib.rlgan_dense_backward_input_split(dz.data_ptr(), n, wb.data_ptr(), dx.data_ptr(), k, m, k, n, slabs.data_ptr(), ns, S())),
         ("weights", k + 1, n, m, lambda ns: lib.rlgan_dense_backward_weights(x.data_ptr(), k, dz.data_ptr(), n, gw.data_ptr(), m, k, n, slabs.data_ptr(), ns, S())))
for name, mo, no, kr, f in prods:
    sug = lib.rlgan_dense_slabs(mo, no, kr)
    out = []
    for ns in (1, 2, 3, 4, 6, 8, 10, 11, 12, 13, 14, 16, 24, 32):
        if ns * mo * no > slabs.numel():
            continue
        t = min(timeit_graph(lambda: f(ns), reps=10) for _ in range(2))
        out.append(f"{ns}{'*' if ns == sug else ''}: {t:6.1f}")
    print(f"{name:8s} m={m} k={k} n={n}  suggested {sug}   us by slabs  " + "  ".join(out))
