"""Oracle (test infrastructure): the replay ring buffers.

Restates `Memory` of GAN-RL/DDPG.py:181-196 (4 fields, `[s, a, r, s_]`) and of
GAN-RL/Curious_GAN.py:155-169 (6 fields, `[s, s_, G, a, r, curious_r]`).
Pinned by `tests/golden/memory_*.npz`, produced by the reference classes themselves.
The index draw is `np.random.choice(capacity, size=n)` on a caller-supplied
`RandomState` (the reference uses the global one).
"""
from __future__ import annotations

import numpy as np


class Memory:
    def __init__(self, capacity: int, dims: int, rng=None):
        self.capacity = capacity
        self.data = np.zeros((capacity, dims))
        self.pointer = 0
        self.rng = rng if rng is not None else np.random

    def store_row(self, row):
        self.data[self.pointer % self.capacity, :] = row
        self.pointer += 1

    def store_transition4(self, s, a, r, s_):          # DDPG.py:187-191
        self.store_row(np.hstack((s, a, [r], s_)))

    def store_transition6(self, s, s_, G, a, r, curious_r):   # Curious_GAN.py:161-165
        self.store_row(np.hstack((s, s_, G, a, [r], curious_r)))

    def sample(self, n: int):
        assert self.pointer >= self.capacity, 'Memory has not been fulfilled'
        idx = self.rng.choice(self.capacity, size=n)
        return self.data[idx, :], idx
