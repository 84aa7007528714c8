#!/usr/bin/env python3
"""Launches one layer's forward (or backward: weight and input gradient) of a pixel encoder a few times at batch B: a
small target for `rocprofv3 --pmc` passes.
usage: convprof.py <doom|universe> <layer> [B] [reps] [fwd|bwd]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl_gan_amd as R

which = sys.argv[1] if len(sys.argv) > 1 else "doom"
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
mode = sys.argv[5] if len(sys.argv) > 5 else "fwd"
hw, spec = (84, R.DOOM_HEAD) if which == "doom" else (42, R.universe_head(2))
enc = R.ConvEncoder(hw, hw, 4, spec, rng=np.random.RandomState(0), bias_init=0.001)
x = torch.rand(B, hw, hw, 4, device="cuda")
enc.forward(x)
L, xin, yout = enc.layers[layer], enc._acts[layer], enc._acts[layer + 1]
dy = torch.randn_like(yout)
torch.cuda.synchronize()
for _ in range(reps):
    if mode == "fwd":
        L.forward(xin)
    else:
        L.backward(xin, yout, dy, need_dx=layer > 0)
torch.cuda.synchronize()
