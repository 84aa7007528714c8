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


@settings(max_examples=25, deadline=None)
@given(m=st.integers(2, 40), n=st.integers(1, 12), seed=st.integers(0, 2**31 - 1), shift=st.floats(-50, 50))
def test_batchnorm_restatement_invariants(m, n, seed, shift):
    """Training-mode batch norm (Conv_model.py:838): the output ignores a constant added to a column, has the
    column mean beta, and its input gradient sums to zero per column (the statistics absorb the mean)."""
    from oracle import inverse_model as om
    rng = np.random.RandomState(seed)
    z = rng.standard_normal((m, n)) * rng.uniform(0.5, 3, n)
    gamma, beta = rng.uniform(0.5, 2, n), rng.uniform(-1, 1, n)
    y, cache = om.bn_forward(z, gamma, beta, True)
    y2, _ = om.bn_forward(z + shift, gamma, beta, True)
    np.testing.assert_allclose(y2, y, rtol=0, atol=1e-8 * (1 + abs(shift)))
    np.testing.assert_allclose(y.mean(0), beta, atol=1e-12)
    dz, dg, db = om.bn_backward(rng.standard_normal((m, n)), cache)
    np.testing.assert_allclose(dz.sum(0), 0, atol=1e-10)
    xh, inv, _ = cache                                   # ... and is orthogonal to the normalised column up to the epsilon term
    np.testing.assert_allclose((dz * xh).sum(0), gamma * inv * dg * (1 - (xh ** 2).mean(0)), atol=1e-9)


@settings(max_examples=25, deadline=None)
@given(m=st.integers(1, 30), n=st.integers(2, 14), seed=st.integers(0, 2**31 - 1), shift=st.floats(-30, 30))
def test_softmax_cross_entropy_invariants(m, n, seed, shift):
    """Sparse softmax cross-entropy against argmax(a) (Conv_model.py:801,818): non-negative, unchanged by a
    constant added to a row of logits, gradient rows sum to zero, and -log(1/n) for equal logits."""
    from oracle import inverse_model as om
    rng = np.random.RandomState(seed)
    logits, a = rng.standard_normal((m, n)) * 3, rng.uniform(0, 1, (m, n))
    loss, d = om.softmax_ce(logits, a)
    loss2, d2 = om.softmax_ce(logits + shift, a)
    assert loss >= 0 and abs(loss - loss2) < 1e-9
    np.testing.assert_allclose(d.sum(1), 0, atol=1e-12)
    np.testing.assert_allclose(d2, d, atol=1e-10)
    assert abs(om.softmax_ce(np.zeros((m, n)), a)[0] - np.log(n)) < 1e-12


@settings(max_examples=20, deadline=None)
@given(n=st.integers(1, 6), u=st.integers(1, 9), seed=st.integers(0, 2**31 - 1))
def test_lstm_step_limits(n, u, seed):
    """BasicLSTMCell (Conv_model.py:85): |h'| < 1; a saturated forget gate with a closed input gate carries the
    cell state through unchanged."""
    from oracle import pixel_ddpg as od
    rng = np.random.RandomState(seed)
    x, c, h = rng.standard_normal((n, u)), rng.standard_normal((n, u)), rng.uniform(-1, 1, (n, u))
    kernel, bias = rng.uniform(-0.5, 0.5, (2 * u, 4 * u)), np.zeros(4 * u)
    c2, h2 = od.lstm_step(x, c, h, kernel, bias)
    assert np.all(np.abs(h2) < 1) and np.all(np.abs(c2) <= np.abs(c) + 1)
    bias[:u], bias[2 * u:3 * u] = -60.0, 60.0            # i closed, f open
    c3, _ = od.lstm_step(x, c, h, kernel * 0, bias)
    np.testing.assert_allclose(c3, c, atol=1e-12)
