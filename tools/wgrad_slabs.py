#!/usr/bin/env python3
"""Weight-gradient time of the universe encoder's layers as a function of the number of slabs (ConvLayer.n_slabs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import rl_gan_amd as R
from kbench import timeit_graph

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
enc = R.ConvEncoder(42, 42, 4, R.universe_head(4), rng=np.random.RandomState(0), bias_init=0.001)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, 42, 42, 4), device="cuda", generator=g)
for li, L in enumerate(enc.layers):
    y = L.forward(x)
    dy = torch.randn(y.shape, device="cuda", generator=g)
    out = []
    for ns in (256, 512, 768, 1024, 1536, 2048, 4096):
        L.n_slabs, L._slabs = ns, None
        L.backward(x, y, dy, need_dx=False)
        t = min(timeit_graph(lambda: L.backward(x, y, dy, need_dx=False), reps=10) for _ in range(2))
        out.append(f"{ns}: {t:6.1f}")
    print(f"layer {li + 1} {tuple(x.shape[1:])} -> {tuple(y.shape[1:])}  us by slabs  " + "  ".join(out))
    x = y
