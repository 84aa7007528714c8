"""Property tests of the CPU restatements (hypothesis): the index draw against NumPy for arbitrary
(population, size, seed), the replay ring's wrap-around, and invariants of the arm environment."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle.arm_env import ScalarArmEnv, VecArmEnv
from oracle.memory import Memory
from oracle.mt19937 import MT19937


@settings(max_examples=40, deadline=None)
@given(n_pop=st.integers(1, 3_000_000), size=st.integers(1, 300), seed=st.integers(0, 2**32 - 1))
def test_choice_is_numpys_for_any_population(n_pop, size, seed):
    """np.random.choice(N, size=n) of the legacy stream (Curious_GAN.py:168), bit for bit, and the
    stream position afterwards (a second draw still agrees)."""
    rs, mt = np.random.RandomState(seed), MT19937(seed)
    np.testing.assert_array_equal(mt.choice(n_pop, size), rs.choice(n_pop, size=size))
    np.testing.assert_array_equal(mt.choice(n_pop, 7), rs.choice(n_pop, size=7))


@settings(max_examples=25, deadline=None)
@given(cap=st.integers(1, 40), n=st.integers(0, 130))
def test_ring_holds_the_last_capacity_rows(cap, n):
    """Memory.store_transition (Curious_GAN.py:161-165): row i lands in slot i % capacity."""
    m = Memory(cap, 25)
    for i in range(n):
        m.store_row(np.full(25, float(i)))
    assert m.pointer == n
    for slot in range(min(cap, n)):
        last = slot + ((n - 1 - slot) // cap) * cap
        assert m.data[slot, 0] == last


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), steps=st.integers(1, 80), mode=st.sampled_from(["easy", "hard"]))
def test_env_invariants_and_vector_form(seed, steps, mode):
    """Angles stay in [0, 2pi), the state is the documented function of ends and goal, rewards are 0 or
    10, `done` only after more than 50 consecutive grabs; the struct-of-arrays form used as the device
    model agrees with the scalar restatement step for step (arm_env.py:43-115)."""
    rng = np.random.RandomState(seed)
    env = ScalarArmEnv(mode, rng=np.random.RandomState(seed))
    vec = VecArmEnv(1, mode)
    u = np.random.RandomState(seed).rand(1, 2)
    s = env.reset()
    sv = vec.reset(np.ones(1, bool), u)
    np.testing.assert_allclose(sv[0], s, atol=1e-12)
    for _ in range(steps):
        a = rng.uniform(-2, 2, 2)
        s, r, done = env.step(a)
        sv, rv, dv = vec.step(a[None])
        np.testing.assert_allclose(sv[0], s, atol=1e-12)
        assert rv[0] == r and bool(dv[0]) == bool(done)
        assert r in (0.0, 10.0) and s[0] in (0.0, 1.0)
        assert np.all(vec.theta >= 0) and np.all(vec.theta < 2 * np.pi)
        assert not done or vec.grab[0] > 50
