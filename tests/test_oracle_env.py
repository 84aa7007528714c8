"""The oracle's ArmEnv restatement against trajectories recorded from the reference module
(tests/golden/arm_env_*.npz, made by tests/golden/gen_fixtures.py from GAN-RL/arm_env.py)."""
import glob
import os

import numpy as np
import pytest

from oracle.arm_env import ScalarArmEnv, VecArmEnv

from conftest import GOLDEN

FILES = sorted(glob.glob(os.path.join(GOLDEN, "arm_env_*_seed*.npz")))


def _mode(path):
    return "hard" if "_hard_" in os.path.basename(path) else "easy"


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_scalar_env_matches_reference_log(path):
    d = np.load(path)
    env = ScalarArmEnv(_mode(path), rng=np.random.RandomState(int(d["np_seed"])))
    for i in range(len(d["is_reset"])):
        if d["is_reset"][i]:
            s = env.reset()
        else:
            s, r, done = env.step(d["action"][i])
            assert r == d["reward"][i]
            assert done == d["done"][i]
        np.testing.assert_array_equal(s, d["state"][i])       # same float64 ops -> bit-exact
        np.testing.assert_array_equal(env.goal, d["goal"][i])
        np.testing.assert_array_equal(env.theta, d["theta"][i])
        assert env.grab_counter == d["grab"][i]


@pytest.mark.parametrize("mode", ["easy", "hard"])
def test_scalar_env_goal_reach(mode):
    d = np.load(os.path.join(GOLDEN, f"arm_env_reach_{mode}.npz"))
    env = ScalarArmEnv(mode, rng=np.random.RandomState(0))
    env.theta[:] = d["theta0"]; env.ends[:] = d["ends0"]; env.goal[:] = d["goal0"]
    rs = np.random.RandomState(int(d["np_seed"]))
    # advance the stream to where the reference's was: one reset (2 doubles) happened before
    rs.rand(2)
    env.rng = rs
    assert d["reward"].tolist().count(10.0) == 1 and d["done"].any()
    for i in range(len(d["is_reset"])):
        if d["is_reset"][i]:
            s = env.reset()
        else:
            s, r, done = env.step(d["action"][i])
            assert r == d["reward"][i] and done == d["done"][i]
        np.testing.assert_array_equal(s, d["state"][i])
        assert env.grab_counter == d["grab"][i]


def test_vec_env_matches_reference_logs():
    """All six logs replayed as a 6-env batch (3 easy + 3 hard run as two batches)."""
    for mode in ("easy", "hard"):
        files = [f for f in FILES if _mode(f) == mode]
        logs = [np.load(f) for f in files]
        n = len(logs)
        env = VecArmEnv(n, mode)
        rss = [np.random.RandomState(int(d["np_seed"])) for d in logs]
        T = len(logs[0]["is_reset"])
        for i in range(T):
            if logs[0]["is_reset"][i]:
                u = np.stack([rs.rand(2) for rs in rss])
                s = env.reset(np.ones(n, bool), u)
            else:
                a = np.stack([d["action"][i] for d in logs])
                s, r, done = env.step(a)
                np.testing.assert_array_equal(r, [d["reward"][i] for d in logs])
                np.testing.assert_array_equal(done, [d["done"][i] for d in logs])
            ref = np.stack([d["state"][i] for d in logs])
            np.testing.assert_allclose(s, ref, rtol=0, atol=1e-13)
            np.testing.assert_array_equal(env.goal, np.stack([d["goal"][i] for d in logs]))


def test_known_answers_of_the_hard_mode_reset_and_first_step():
    """arm_env.py:43-78,91-115 with np.random.seed(1): the goal the reset draws (truncated to integers) and the state
    after one step of a = (0.5, -2.0), worked out by hand from the reference's formulas."""
    np.random.seed(1)
    env = ScalarArmEnv("hard")
    s = env.reset()
    np.testing.assert_allclose(s, [0, -0.83, -1.44, -0.83, -1.44, 0.17, -0.44], atol=1e-12)
    assert env.goal.tolist() == [166, 288]
    s, r, d = env.step(np.array([0.5, -2.0]))
    np.testing.assert_allclose(s, [0, 0.66937513, -0.41501042, 1.16687721, -0.46492712, 0.17, -0.44], atol=1e-8)
    np.testing.assert_allclose(env.theta, [0.05, 6.18318531], atol=1e-8)
    assert r == 0 and d is False
