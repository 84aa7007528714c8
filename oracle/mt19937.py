"""Oracle (test infrastructure): MT19937 and the legacy NumPy draws built on it.

`np.random.choice(N, size=n)` on the legacy `RandomState` (used by `Memory.sample`,
GAN-RL/Curious_GAN.py:168, DDPG.py:195) is `randint(0, N, size=n)`, which for
N-1 <= 0xFFFFFFFF draws 32-bit outputs, masks them with the smallest 2^k-1 >= N-1 and
rejects values > N-1, in stream order.  Restated here from the published MT19937
algorithm (Matsumoto & Nishimura 1998) and NumPy's documented legacy behaviour; pinned
by comparison with NumPy's own `RandomState` in `tests/test_oracle_sampler.py`.

Also restated: `random_sample` (53-bit doubles, used by `rand` in arm_env.py:66,69) and
the legacy polar-method gaussian (used by `np.random.normal`, Curious_GAN.py:191).
"""
from __future__ import annotations

import numpy as np

N, M = 624, 397
MATRIX_A, UPPER, LOWER = 0x9908B0DF, 0x80000000, 0x7FFFFFFF


class MT19937:
    def __init__(self, seed: int | None = None):
        self.mt = np.zeros(N, dtype=np.uint32)
        self.pos = N
        self.has_gauss, self.gauss = 0, 0.0
        if seed is not None:
            self.seed(seed)

    def seed(self, seed: int):
        """init_genrand -- what `RandomState(int)` / `np.random.seed(int)` do."""
        mt = [0] * N
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt = np.array(mt, dtype=np.uint32)
        self.pos = N
        self.has_gauss, self.gauss = 0, 0.0

    @classmethod
    def from_numpy(cls, rs: np.random.RandomState) -> "MT19937":
        name, key, pos, has_gauss, cached = rs.get_state()
        assert name == "MT19937"
        g = cls()
        g.mt, g.pos, g.has_gauss, g.gauss = key.astype(np.uint32).copy(), int(pos), int(has_gauss), float(cached)
        return g

    def to_numpy_state(self):
        return ("MT19937", self.mt.copy(), self.pos, self.has_gauss, self.gauss)

    def _twist(self):
        mt = [int(x) for x in self.mt]
        for k in range(N):
            y = (mt[k] & UPPER) | (mt[(k + 1) % N] & LOWER)
            mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (MATRIX_A if (y & 1) else 0)
        self.mt = np.array(mt, dtype=np.uint32)
        self.pos = 0

    def next_uint32(self) -> int:
        if self.pos >= N:
            self._twist()
        y = int(self.mt[self.pos])
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def next_double(self) -> float:
        a, b = self.next_uint32() >> 5, self.next_uint32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    def choice(self, n_pop: int, size: int) -> np.ndarray:
        """Indices of `RandomState.choice(n_pop, size=size)` (uniform, with replacement)."""
        rng = n_pop - 1
        out = np.zeros(size, dtype=np.int64)
        if rng == 0:
            return out
        assert rng <= 0xFFFFFFFF
        mask = rng
        for sh in (1, 2, 4, 8, 16):
            mask |= mask >> sh
        i = 0
        while i < size:
            v = self.next_uint32() & mask
            if v <= rng:
                out[i] = v
                i += 1
        return out

    def legacy_gauss(self) -> float:
        """Marsaglia polar method with one cached value (NumPy legacy_gauss)."""
        if self.has_gauss:
            self.has_gauss = 0
            t, self.gauss = self.gauss, 0.0
            return t
        while True:
            x1 = 2.0 * self.next_double() - 1.0
            x2 = 2.0 * self.next_double() - 1.0
            r2 = x1 * x1 + x2 * x2
            if 0.0 < r2 < 1.0:
                break
        f = np.sqrt(-2.0 * np.log(r2) / r2)
        self.gauss, self.has_gauss = f * x1, 1
        return f * x2

    def normal(self, loc, scale):
        loc = np.atleast_1d(np.asarray(loc, dtype=np.float64))
        return np.array([l + scale * self.legacy_gauss() for l in loc])
