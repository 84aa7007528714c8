#!/usr/bin/env python3
"""A few launches of the forward kernel at B rows (default 32768 = 8 tiles per CU, the throughput
regime): a target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_gan_amd import _lib
from rl_gan_amd.core import DeviceNet, mlp_forward
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
net = DeviceNet(_lib.NET_CRITIC, 7, 2, trainable=False, rng=np.random.RandomState(0))
s = torch.rand(B, 7, device="cuda"); a = torch.rand(B, 2, device="cuda")
for _ in range(5):
    mlp_forward(net, s, a=a)
torch.cuda.synchronize()
