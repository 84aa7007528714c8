#!/usr/bin/env python3
"""Forward time of the encoders' FIRST layers at batch B, from float32 frames and from the uint8 ring in place, with a
checksum of the output (two library builds must print the same sums): a short target for A/B runs.
  python tools/firstlayer.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import rl_gan_amd as R
from rl_gan_amd import FrameRef
from kbench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, hw, spec in (("doom 84", 84, R.DOOM_HEAD), ("universe 84", 84, R.universe_head(2)), ("universe 42", 42, R.universe_head(2))):
    enc = R.ConvEncoder(hw, hw, 4, spec, rng=np.random.RandomState(0), bias_init=0.001)
    L = enc.layers[0]
    g = torch.Generator(device="cuda").manual_seed(1)
    ring = torch.randint(0, 256, (B + 3, hw * hw * 4), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randperm(B + 3, device="cuda", generator=g)[:B].contiguous()
    x = (ring[idx].to(torch.float32) * (1.0 / 255.0)).reshape(B, hw, hw, 4).contiguous()
    ref = FrameRef(ring, idx, 1.0 / 255.0, B)
    y = L.forward(x)
    yu = L.forward_frames(ref)
    tf = min(timeit(lambda: L.forward(x)) for _ in range(3))
    tu = min(timeit(lambda: L.forward_frames(ref)) for _ in range(3))
    gb = (x.numel() * 4 + y.numel() * 4) / 1e9
    print(f"{name}: f32 {tf:7.1f} us ({gb / tf * 1e3:5.2f} TB/s)   u8 ring {tu:7.1f} us   sum {y.double().sum().item():.6f} "
          f"abs {y.double().abs().sum().item():.6f}  u8-f32 max|d| {(y - yu).abs().max().item():.2e}")
