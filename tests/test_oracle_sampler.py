"""MT19937 / Philox oracle against NumPy's own legacy stream, the committed index fixtures and
the Philox paper's known-answer vectors."""
import os

import numpy as np
import pytest

from oracle import mt19937, philox

from conftest import GOLDEN


@pytest.mark.parametrize("N,n,seed", [(5000, 16, 1), (5000, 32, 1), (5000, 4096, 1), (2 ** 20, 4096, 7), (1000003, 4096, 3)])
def test_choice_matches_numpy_and_fixture(N, n, seed):
    d = np.load(os.path.join(GOLDEN, "sampler_indices.npz"))
    got = mt19937.MT19937(seed).choice(N, n)
    np.testing.assert_array_equal(got, np.random.RandomState(seed).choice(N, size=n))
    np.testing.assert_array_equal(got, d[f"idx_{N}_{n}_{seed}"])


def test_known_answer_first_six():
    assert mt19937.MT19937(1).choice(5000, 4096)[:6].tolist() == [235, 3980, 905, 2763, 2895, 144]


def test_stream_continuity_and_state_roundtrip():
    d = np.load(os.path.join(GOLDEN, "sampler_indices.npz"))
    g = mt19937.MT19937(9)
    np.testing.assert_array_equal(g.choice(5000, 4096), d["idx_5000_4096_9_first"])
    np.testing.assert_array_equal(g.choice(5000, 4096), d["idx_5000_4096_9_second"])
    rs = np.random.RandomState(9)
    rs.choice(5000, size=100)
    g2 = mt19937.MT19937.from_numpy(rs)
    np.testing.assert_array_equal(g2.choice(777, 50), rs.choice(777, size=50))
    rs2 = np.random.RandomState(0)
    rs2.set_state(g2.to_numpy_state())
    assert rs2.randint(0, 2 ** 31) == rs.randint(0, 2 ** 31)


def test_doubles_and_legacy_gauss():
    g, rs = mt19937.MT19937(5), np.random.RandomState(5)
    assert [g.next_double() for _ in range(10)] == rs.rand(10).tolist()
    for _ in range(7):
        np.testing.assert_array_equal(g.normal([0.3, -0.2], 2.0), rs.normal([0.3, -0.2], 2.0))


def test_edge_population_sizes():
    for N in (1, 2, 3, 4, 5, 255, 256, 257, 65536):
        np.testing.assert_array_equal(mt19937.MT19937(3).choice(N, 64), np.random.RandomState(3).choice(N, size=64))


def test_philox_known_answers():
    assert philox.philox4x32((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox.philox4x32((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert philox.philox4x32((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_philox_normal_statistics():
    z = np.array([philox.normal_pair(i, 0, 42) for i in range(4000)]).ravel()
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05
    u = np.array([philox.uniform_pair(i, 1, 42) for i in range(2000)]).ravel()
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.03
