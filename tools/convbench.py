#!/usr/bin/env python3
"""Throughput of the pixel encoders at batch B (forward, and forward+backward), per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rl_gan_amd as R
from kbench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, hw, spec in (("doom 84x84x4", 84, R.DOOM_HEAD), ("universe(2) 84x84x4", 84, R.universe_head(2)), ("universe(4) 42x42x4", 42, R.universe_head(4)),
                       ("universe(2) 42x42x4", 42, R.universe_head(2))):
    enc = R.ConvEncoder(hw, hw, 4, spec, rng=np.random.RandomState(0), bias_init=0.001)
    x = torch.rand(B, hw, hw, 4, device="cuda")
    feat = enc.forward(x)
    d = torch.randn_like(feat)
    macs = sum(L.h_out * L.w_out * L.k * L.k * L.c_in * L.c_out for L in enc.layers)
    tf = timeit(lambda: enc.forward(x))
    tb = timeit(lambda: enc.backward(d))
    print(f"{name:22s} B={B}: forward {tf:8.1f} us ({2 * macs * B / tf / 1e6:6.1f} TFLOP/s, input {x.numel() * 4 / tf / 1e6:6.2f} TB/s)   "
          f"backward {tb:8.1f} us ({4 * macs * B / tb / 1e6:6.1f} TFLOP/s)")
    acts = enc._acts
    for i, L in enumerate(enc.layers):
        m = L.h_out * L.w_out * L.k * L.k * L.c_in * L.c_out
        t1 = timeit(lambda: L.forward(acts[i]))
        dy = torch.randn_like(acts[i + 1])
        t2 = timeit(lambda: L.backward(acts[i], acts[i + 1], dy, need_dx=i > 0))
        print(f"    layer {i}: {L.conv.h}x{L.conv.w}x{L.c_in} -> {L.h_out}x{L.w_out}x{L.c_out}  fwd {t1:8.1f} us {2 * m * B / t1 / 1e6:6.1f} TF/s   bwd {t2:8.1f} us")
    # first layer reading uint8 frames in place from a replay ring through an index array
    mem = R.FrameMemory(3 * B, (hw, hw, 4), rng=R.MT19937Device(1))
    for j in range(3):
        mem.store_transitions((x * 255).to(torch.uint8), (x * 255).to(torch.uint8))
    ref = mem.ref(mem.sample_indices(B))
    L, y1 = enc.layers[0], acts[1]
    dy = torch.randn_like(y1)
    t0 = timeit(lambda: mem.gather(ref.idx))
    t1 = timeit(lambda: L.forward_frames(ref))
    t2 = timeit(lambda: L.backward_frames(ref, y1, dy))
    print(f"    layer 0 on uint8 ring: gather alone {t0:8.1f} us; fwd in place {t1:8.1f} us; weight gradient in place {t2:8.1f} us")

for k in (64, 2592, 12800):
    L = R.DenseLayer(k, 256, rng=np.random.RandomState(0))
    x = torch.randn(B, k, device="cuda")
    y = L.forward(x)
    dy = torch.randn_like(y)
    t1 = timeit(lambda: L.forward(x))
    t2 = timeit(lambda: L.backward(x, y, dy))
    fl = 2.0 * B * (k + 1) * 256
    print(f"dense {k:6d} -> 256  B={B}: forward {t1:8.1f} us ({fl / t1 / 1e6:6.1f} TFLOP/s)   backward (dW + dX) {t2:8.1f} us ({2 * fl / t2 / 1e6:6.1f} TFLOP/s)")
