#!/usr/bin/env python3
"""Throughput of the pixel generator's decoder (1x1x200 -> 42x42x1) at batch B, per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rl_gan_amd as R
from kbench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dec = R.Decoder(200, rng=np.random.RandomState(0), bias_init=0.001)
code = torch.rand(B, 200, device="cuda")
img = dec.forward(code)
d = torch.randn_like(img)
macs = sum(L.h * L.w * L.c_in * L.kk for L in dec.layers)
tf = timeit(lambda: dec.forward(code))
tb = timeit(lambda: dec.backward(d))
print(f"decoder B={B}: forward {tf:8.1f} us ({2 * macs * B / tf / 1e6:6.1f} TFLOP/s)   backward {tb:8.1f} us ({4 * macs * B / tb / 1e6:6.1f} TFLOP/s)")
acts = dec._acts
for i, L in enumerate(dec.layers):
    m = L.h * L.w * L.c_in * L.kk
    last = i + 1 == len(dec.layers)
    t1 = timeit(lambda: L.forward(acts[i], act=not last))
    dy = torch.randn_like(acts[i + 1])
    t2 = timeit(lambda: L.backward(acts[i], acts[i + 1], dy, act=not last))
    cols_mb = B * L.h * L.w * L.kk * 4 / 1e6
    print(f"    layer {i}: {L.h}x{L.w}x{L.c_in} -> {L.h_out}x{L.w_out}x{L.c_out}  fwd {t1:8.1f} us {2 * m * B / t1 / 1e6:6.1f} TF/s   bwd {t2:8.1f} us {4 * m * B / t2 / 1e6:6.1f} TF/s   (cols {cols_mb:7.1f} MB)")
