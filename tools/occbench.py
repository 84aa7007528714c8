#!/usr/bin/env python3
"""Forward-pass time of one network versus the number of 16-row tiles per compute unit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rl_gan_amd as R
from rl_gan_amd import _lib
from rl_gan_amd.core import DeviceNet, mlp_forward
from kbench import timeit

net = DeviceNet(_lib.NET_CRITIC, 7, 2, trainable=False, rng=np.random.RandomState(0))
for B in (2048, 4096, 8192, 12288, 16384, 24576, 32768):
    s = torch.rand(B, 7, device="cuda"); a = torch.rand(B, 2, device="cuda")
    us = timeit(lambda: mlp_forward(net, s, a=a))
    print(f"B={B:6d} tiles/CU={B / 16 / 256:5.2f}  {us:7.1f} us   {us / (B / 4096):6.2f} us per 4096 rows   {2 * 43810 * B / us / 1e6:6.1f} TFLOP/s")
