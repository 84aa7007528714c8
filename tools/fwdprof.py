#!/usr/bin/env python3
"""A few forward passes of the critic at batch B: a small target for rocprofv3 --pmc.  usage: fwdprof.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_gan_amd import _lib
from rl_gan_amd.core import DeviceNet, mlp_forward
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
net = DeviceNet(_lib.NET_CRITIC, 7, 2, trainable=False, rng=np.random.RandomState(0))
s = torch.rand(B, 7, device="cuda"); a = torch.rand(B, 2, device="cuda")
torch.cuda.synchronize()
for _ in range(4):
    mlp_forward(net, s, a=a)
torch.cuda.synchronize()
