#!/usr/bin/env python3
"""Backward of one conv layer split into its launches: ELU', weight gradient (+ slab sum), input gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rl_gan_amd as R
from rl_gan_amd._lib import lib, check
from rl_gan_amd.core import _stream
from kbench import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for (h, cin, cout, k, s) in ((84, 4, 32, 3, 2), (41, 32, 32, 3, 2), (20, 32, 32, 3, 2), (84, 4, 8, 5, 4), (20, 8, 16, 3, 2)):
    L = R.ConvLayer(h, h, cin, cout, k, s, rng=np.random.RandomState(0))
    if NS: L.n_slabs = NS
    x = torch.rand(B, h, h, cin, device="cuda")
    y = L.forward(x)
    dy = torch.randn_like(y)
    dz = torch.empty_like(dy)
    L.backward(x, y, dy)
    dx = torch.empty_like(x)
    t0 = timeit(lambda: check(lib.rlgan_elu_backward(dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), _stream())))
    sl, ns = L._slab_buffer(B)
    t1 = timeit(lambda: check(lib.rlgan_conv_backward_weights(L.conv, x.data_ptr(), dz.data_ptr(), None, sl.data_ptr(), ns,
                                                              L.grad.data_ptr(), B, _stream())))
    t2 = timeit(lambda: check(lib.rlgan_conv_backward_input(L.conv, dz.data_ptr(), None, L.canon.data_ptr(), None, dx.data_ptr(), B, _stream())))
    t3 = timeit(lambda: check(lib.rlgan_conv_backward_input(L.conv, dz.data_ptr(), None, L.canon.data_ptr(), L.wdx.data_ptr(), dx.data_ptr(), B, _stream())))
    macs = L.h_out * L.w_out * k * k * cin * cout * B
    print(f"{h}x{h}x{cin} -> {L.h_out}x{L.w_out}x{cout} k{k}/s{s}: elu' {t0:7.1f} us  dW {t1:8.1f} us ({2 * macs / t1 / 1e6:5.1f} TF/s)  dX gather {t2:8.1f} us ({2 * macs / t2 / 1e6:5.1f} TF/s)  dX mfma {t3:8.1f} us ({2 * macs / t3 / 1e6:5.1f} TF/s)")
