#!/usr/bin/env python3
"""Times the uint8 frame gather (and store) at the pixel variants' shape: 84x84x4 stacks, minibatch 4096.
Algorithmic bytes per gathered frame: frame_bytes read + 4*frame_bytes written."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_gan_amd as R

cap, shape, B = 20000, (84, 84, 4), 4096
m = R.FrameMemory(cap, shape, rng=R.MT19937Device(1))
g = torch.Generator(device="cuda"); g.manual_seed(0)
f = torch.randint(0, 256, (B,) + shape, dtype=torch.uint8, device="cuda", generator=g)
for i in range(5):
    m.store_transitions(f, f)
idx = m.sample_indices(B)
out = torch.empty((B,) + shape, dtype=torch.float32, device="cuda")
for name, fn, nbytes in (("gather", lambda: m.gather(idx, "s", out), 5 * B * m.frame_bytes),
                         ("store ", lambda: m.store_transitions(f, f), 4 * B * m.frame_bytes)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"{name}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s")
