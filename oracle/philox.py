"""Oracle (test infrastructure): Philox4x32-10 counter RNG and the draws built on it.

The reference has no counter-based generator: its exploration noise and goal draws come
from NumPy's global MT19937 (Curious_GAN.py:191, arm_env.py:66).  Throughput runs over
thousands of environments need an order-independent stream instead, so the device uses
Philox4x32-10 (Salmon et al., SC'11, "Parallel random numbers: as easy as 1, 2, 3");
this file restates it so the device kernel can be checked word for word.  Pinned by the
paper's known-answer vectors in `tests/test_oracle_sampler.py`.

Stream layout used by the device kernels (and restated here):
  counter = (index_lo, index_hi, stream, 0), key = (seed_lo, seed_hi)
  normal pair  : Box-Muller on words 0,1 -> (z0, z1);  u = (w + 0.5) * 2^-32 in (0,1)
  uniform pair : words 0..3 -> two 53-bit doubles, ((w0>>5)*2^26 + (w1>>6)) * 2^-53
"""
from __future__ import annotations

import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32(counter, key, rounds: int = 10):
    c0, c1, c2, c3 = [int(c) & MASK for c in counter]
    k0, k1 = [int(k) & MASK for k in key]
    for _ in range(rounds):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> 32, p0 & MASK, p1 >> 32, p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK, lo1, (hi0 ^ c3 ^ k1) & MASK, lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def _words(index: int, stream: int, seed: int):
    return philox4x32((index & MASK, (index >> 32) & MASK, stream & MASK, 0),
                      (seed & MASK, (seed >> 32) & MASK))


def normal_pair(index: int, stream: int, seed: int):
    """float32 Box-Muller pair, computed in float32 like the device."""
    w = _words(index, stream, seed)
    u0 = np.float32((np.float32(w[0]) + np.float32(0.5)) * np.float32(2.0 ** -32))
    u1 = np.float32((np.float32(w[1]) + np.float32(0.5)) * np.float32(2.0 ** -32))
    # float32(w)+0.5 can round up to 2^32 -> u == 1 exactly; the device clamps below 1
    u0 = min(u0, np.float32(1.0 - 2.0 ** -24))
    r = np.sqrt(np.float32(-2.0) * np.log(u0, dtype=np.float32), dtype=np.float32)
    ang = np.float32(2.0 * np.pi) * u1
    return np.float32(r * np.cos(ang, dtype=np.float32)), np.float32(r * np.sin(ang, dtype=np.float32))


def uniform_pair(index: int, stream: int, seed: int):
    w = _words(index, stream, seed)
    a = ((w[0] >> 5) * 67108864.0 + (w[1] >> 6)) / 9007199254740992.0
    b = ((w[2] >> 5) * 67108864.0 + (w[3] >> 6)) / 9007199254740992.0
    return a, b
