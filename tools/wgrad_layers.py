#!/usr/bin/env python3
"""Weight-gradient and input-gradient time of every layer of the universe encoder (42x42 frames) at batch B, with
checksums (two library builds must print the same sums up to summation order): a short target for A/B runs.
  python tools/wgrad_layers.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import rl_gan_amd as R
from kbench import timeit, timeit_graph

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
enc = R.ConvEncoder(42, 42, 4, R.universe_head(4), rng=np.random.RandomState(0), bias_init=0.001)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, 42, 42, 4), device="cuda", generator=g)
for li, L in enumerate(enc.layers):
    y = L.forward(x)
    dy = torch.randn(y.shape, device="cuda", generator=g)
    L.backward(x, y, dy, need_dx=False)
    t = min(timeit(lambda: L.backward(x, y, dy, need_dx=False)) for _ in range(3))
    gr = L.grad.double()
    flops = 2.0 * y.numel() * (L.k * L.k * L.c_in + 1)
    print(f"layer {li + 1}: in {tuple(x.shape[1:])} out {tuple(y.shape[1:])}  wgrad {t:7.1f} us ({flops / t * 1e-6:6.1f} TFLOP/s)  "
          f"sum {gr.sum().item():.6f} abs {gr.abs().sum().item():.6f}")
    if True:                     # the first layer's input gradient too: the generator's image is its input
        dx = L.backward(x, y, dy, need_dx=True, need_dw=False)
        td = min(timeit_graph(lambda: L.backward(x, y, dy, need_dx=True, need_dw=False), reps=10) for _ in range(3))
        print(f"         dx {td:7.1f} us ({flops / td * 1e-6:6.1f} TFLOP/s)  sum {dx.double().sum().item():.6f} abs {dx.double().abs().sum().item():.6f}")
    x = y
