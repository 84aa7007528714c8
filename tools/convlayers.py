#!/usr/bin/env python3
"""Forward time of every layer of the pixel encoders at batch B (float32 frames): a short target for A/B runs."""
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
for name, hw, spec in (("doom", 84, R.DOOM_HEAD), ("universe(4) 42", 42, R.universe_head(4))):
    enc = R.ConvEncoder(hw, hw, 4, spec, rng=np.random.RandomState(0), bias_init=0.001)
    x = torch.rand(B, hw, hw, 4, device="cuda")
    enc.forward(x)
    acts = enc._acts
    for i, L in enumerate(enc.layers):
        out = torch.empty_like(acts[i + 1])
        t = min(timeit_graph(lambda: L.forward(acts[i], out=out)) for _ in range(3))
        print(f"{name} layer {i}: fwd {t:8.1f} us")
