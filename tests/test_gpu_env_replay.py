"""Device ArmEnv, replay ring, MT19937 sampler and Philox draws against the golden vectors recorded
from the reference (tests/golden/*.npz) and against the oracle.
Reference: arm_env.py:43-78,91-115; Curious_GAN.py:155-169; DDPG.py:181-196."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

# float64 device cos/sin/fmod may differ from the host libm in the last bits; states are O(1)
ENV_ATOL = 1e-12


def _mode(path):
    return "hard" if "_hard_" in os.path.basename(path) else "easy"


def test_scalar_armenv_replays_reference_logs():
    import rl_gan_amd as R
    for path in sorted(glob.glob(os.path.join(GOLDEN, "arm_env_*_seed*.npz"))):
        d = np.load(path)
        np.random.seed(int(d["np_seed"]))            # the drop-in draws from the global stream like the reference
        env = R.ArmEnv(mode=_mode(path))
        for i in range(len(d["is_reset"])):
            if d["is_reset"][i]:
                s = env.reset()
            else:
                s, r, done = env.step(d["action"][i])
                assert r == d["reward"][i] and done == bool(d["done"][i])
            np.testing.assert_allclose(s, d["state"][i], rtol=0, atol=ENV_ATOL)
            np.testing.assert_array_equal(env.point_info, d["goal"][i])
            np.testing.assert_allclose(env.arm_info[:, 1], d["theta"][i], rtol=0, atol=ENV_ATOL)
            assert env.grab_counter == int(d["grab"][i])
        assert env.state_dim == 7 and env.action_dim == 2 and env.action_bound == [-1, 1]


@pytest.mark.parametrize("mode", ["easy", "hard"])
def test_goal_reach_episode(mode):
    import rl_gan_amd as R
    d = np.load(os.path.join(GOLDEN, f"arm_env_reach_{mode}.npz"))
    np.random.seed(int(d["np_seed"])); np.random.rand(2)
    env = R.ArmEnv(mode=mode)
    v = env._v
    v.theta.copy_(torch.from_numpy(d["theta0"])); v.ends.copy_(torch.from_numpy(d["ends0"].reshape(-1)))
    v.goal.copy_(torch.from_numpy(d["goal0"].astype(np.float64)))
    for i in range(len(d["is_reset"])):
        if d["is_reset"][i]:
            s = env.reset()
        else:
            s, r, done = env.step(d["action"][i])
            assert r == d["reward"][i] and done == bool(d["done"][i])
        np.testing.assert_allclose(s, d["state"][i], rtol=0, atol=ENV_ATOL)
        assert env.grab_counter == int(d["grab"][i])
    assert d["reward"].tolist().count(10.0) == 1


@pytest.mark.parametrize("mode", ["easy", "hard"])
def test_vec_env_4096_against_oracle(mode):
    import rl_gan_amd as R
    from oracle.arm_env import VecArmEnv as OEnv
    n = 4096
    rng = np.random.RandomState(3)
    env, o = R.VecArmEnv(n, mode), OEnv(n, mode)
    for t in range(60):
        if t % 25 == 0:
            u = rng.rand(n, 2)
            mask = np.ones(n, bool) if t == 0 else rng.rand(n) < 0.3
            so = o.reset(mask, u)
            sd = env.reset(torch.from_numpy(u).cuda(), torch.from_numpy(mask.astype(np.uint8)).cuda()).cpu().numpy()
            np.testing.assert_allclose(sd[mask], so[mask], rtol=0, atol=ENV_ATOL)
        a = rng.uniform(-1.5, 1.5, (n, 2))
        if t > 30:
            a *= 0.0        # hold still: lets grab counters run
        so, ro, do = o.step(a)
        sd, rd, dd = env.step(torch.from_numpy(a).cuda())
        np.testing.assert_allclose(sd.cpu().numpy(), so, rtol=0, atol=ENV_ATOL)
        np.testing.assert_array_equal(rd.cpu().numpy(), ro)
        np.testing.assert_array_equal(dd.cpu().numpy().astype(bool), do)
        np.testing.assert_array_equal(env.grab.cpu().numpy(), o.grab)
        np.testing.assert_array_equal(env.goal.cpu().numpy().reshape(n, 2), o.goal)
        np.testing.assert_allclose(env.state32.cpu().numpy(), so.astype(np.float32), rtol=0, atol=1e-6)
    th = env.theta.cpu().numpy()
    assert (th >= 0).all() and (th < 2 * np.pi).all()


def test_memory_ring_layout_wrap_and_sample_match_reference():
    import rl_gan_amd as R
    d = np.load(os.path.join(GOLDEN, "memory_ring.npz"))
    cap = int(d["capacity"])
    m4 = R.DDPGMemory(cap, 17)
    m6 = R.Memory(cap, 25)
    with pytest.raises(AssertionError, match="Memory has not been fulfilled"):
        m6.sample(4)
    for row in d["rows4"]:
        m4.store_transition(row[:7], row[7:9], row[9], row[10:])
    for row in d["rows6"]:
        m6.store_transition(row[:7], row[7:14], row[14:21], row[21:23], row[23], row[24:])
    assert m4.pointer == int(d["pointer4"]) and m6.pointer == int(d["pointer6"])
    np.testing.assert_array_equal(m4.data, d["data4"].astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(m6.data, d["data6"].astype(np.float32).astype(np.float64))
    np.random.seed(int(d["sample_seed4"]))
    np.testing.assert_array_equal(m4.sample(16), d["batch4"].astype(np.float32).astype(np.float64))
    after = np.random.rand()
    np.random.seed(int(d["sample_seed4"])); np.random.choice(cap, size=16)
    assert after == np.random.rand()                  # the global stream advanced exactly like NumPy's
    np.random.seed(int(d["sample_seed6"]))
    np.testing.assert_array_equal(m6.sample(16), d["batch6"].astype(np.float32).astype(np.float64))


def test_sampler_bit_exact_against_fixtures_and_numpy():
    import rl_gan_amd as R
    d = np.load(os.path.join(GOLDEN, "sampler_indices.npz"))
    for (N, n, seed) in [(5000, 16, 1), (5000, 32, 1), (5000, 4096, 1), (2 ** 20, 4096, 7), (1000003, 4096, 3)]:
        g = R.MT19937Device(seed)
        idx = g.choice(N, n).cpu().numpy()
        assert idx.dtype == np.int64
        np.testing.assert_array_equal(idx, d[f"idx_{N}_{n}_{seed}"])
    g = R.MT19937Device(9)
    np.testing.assert_array_equal(g.choice(5000, 4096).cpu().numpy(), d["idx_5000_4096_9_first"])
    np.testing.assert_array_equal(g.choice(5000, 4096).cpu().numpy(), d["idx_5000_4096_9_second"])
    assert g.choice(5000, 4096)[:0].numel() == 0
    # edge populations, odd sizes, state hand-over from and to NumPy
    rs = np.random.RandomState(21)
    rs.rand(5)
    g.set_numpy_state(rs.get_state())
    for N, n in [(1, 7), (2, 33), (3, 100), (255, 64), (256, 64), (257, 1000), (65536, 625), (2 ** 31, 50), (2 ** 32, 50)]:
        np.testing.assert_array_equal(g.choice(N, n).cpu().numpy(), rs.choice(N, size=n) if N < 2 ** 31 else rs.randint(0, N, size=n, dtype=np.int64))
    rs2 = np.random.RandomState(0)
    rs2.set_state(g.get_numpy_state())
    assert rs2.rand() == rs.rand()


def test_gather_full_size_properties_and_bad_index():
    import rl_gan_amd as R
    cap, n = 1 << 20, 4096
    m = R.DeviceMemory(cap, 25, rng=R.MT19937Device(5))
    rows = torch.arange(cap, dtype=torch.float32, device="cuda").reshape(-1, 1).repeat(1, 25)
    rows += torch.arange(25, dtype=torch.float32, device="cuda") / 32.0
    for k in range(0, cap, 1 << 18):
        m.store_rows(rows[k:k + (1 << 18)])
    m.store_rows(rows[:1000] + 0.5)                       # wrap: slots 0..999 overwritten
    assert m.pointer == cap + 1000
    idx = m.sample_indices(n)
    out = m.gather(idx)
    ref = np.random.RandomState(5).choice(cap, size=n)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    exp = ref[:, None].astype(np.float64) + np.arange(25) / 32.0 + np.where(ref < 1000, 0.5, 0.0)[:, None]
    np.testing.assert_array_equal(out.cpu().numpy().astype(np.float64), exp.astype(np.float32).astype(np.float64))
    bad = torch.tensor([0, cap, -1, 5], dtype=torch.int64, device="cuda")
    o = m.gather(bad).cpu().numpy()
    assert np.isnan(o[1]).all() and np.isnan(o[2]).all() and not np.isnan(o[0]).any()


def test_philox_words_and_draws_match_oracle():
    from rl_gan_amd import _lib
    from rl_gan_amd.core import _stream
    from oracle import philox
    n = 512
    z = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    u = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
    seed, i0 = (77 << 32) | 1234, (1 << 33) + 5
    _lib.check(_lib.lib.rlgan_philox_normal(z.data_ptr(), n, seed, 3, i0, None, 0, _stream()))
    _lib.check(_lib.lib.rlgan_philox_uniform(u.data_ptr(), n, seed, 4, i0, None, 0, _stream()))
    zr = np.array([philox.normal_pair(i0 + i, 3, seed) for i in range(n)])
    ur = np.array([philox.uniform_pair(i0 + i, 4, seed) for i in range(n)])
    np.testing.assert_array_equal(u.cpu().numpy(), ur)              # integer path: bit-exact
    np.testing.assert_allclose(z.cpu().numpy(), zr, rtol=2e-5, atol=2e-6)   # float32 libm differences only
    big = torch.zeros((1 << 16, 2), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib.rlgan_philox_normal(big.data_ptr(), 1 << 16, 9, 0, 0, None, 0, _stream()))
    b = big.cpu().numpy()
    assert np.isfinite(b).all() and abs(b.mean()) < 0.01 and abs(b.std() - 1) < 0.01
