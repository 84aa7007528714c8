"""Generate the golden vectors under tests/golden/ from the reference's own NumPy code.

Run ONLY in the build container (it reads /root/reference, which never travels):

    python tests/golden/gen_fixtures.py

What it does: imports `GAN-RL/arm_env.py`, `GAN-RL/DDPG.py` and `GAN-RL/Curious_GAN.py`
with inert stand-ins for the third-party modules that are absent here (`pyglet`,
`tensorflow`; only referenced at import time / inside TF methods that are never called),
drives `ArmEnv` and the two `Memory` classes with seeded inputs and stores inputs and
outputs as small .npz files.  The files hold data only -- no reference source text.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/GAN-RL"


def _install_stand_ins():
    sys.dont_write_bytecode = True
    pyglet = types.ModuleType("pyglet")
    pyglet.clock = types.SimpleNamespace(set_fps_limit=lambda *a, **k: None, ClockDisplay=lambda *a, **k: None)
    pyglet.window = types.SimpleNamespace(Window=object)
    pyglet.gl = types.SimpleNamespace()
    pyglet.graphics = types.SimpleNamespace(Batch=object)
    sys.modules["pyglet"] = pyglet
    sys.modules["tensorflow"] = types.ModuleType("tensorflow")
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)


def gen_arm_env(arm_env):
    """Event logs.  Each event is a reset (is_reset=1, action=nan) or a step; after every event
    the returned state, reward (nan for a reset), done flag, goal and joint angles are recorded.
    3 seeds x 2 modes x 300 steps with a reset every 100 steps, plus one scripted episode per
    mode that parks the finger on the goal for more than 50 steps and then leaves."""
    def run(env, events):
        log = {k: [] for k in ("is_reset", "action", "state", "reward", "done", "goal", "theta", "grab")}
        for ev in events:
            if ev is None:
                s = env.reset(); r = np.nan; d = False; a = np.full(2, np.nan)
            else:
                a = np.asarray(ev, dtype=np.float64)
                s, r, d = env.step(a)
            log["is_reset"].append(ev is None); log["action"].append(a); log["state"].append(np.asarray(s, dtype=np.float64))
            log["reward"].append(r); log["done"].append(bool(d)); log["goal"].append(env.point_info.copy())
            log["theta"].append(env.arm_info[:, 1].copy()); log["grab"].append(env.grab_counter)
        return {k: np.array(v) for k, v in log.items()}

    for mode in ("easy", "hard"):
        for seed in (1, 2, 3):
            np.random.seed(seed)
            env = arm_env.ArmEnv(mode=mode)
            act_rng = np.random.RandomState(1000 + seed)
            events = []
            for ep in range(3):
                events.append(None)
                events += [act_rng.uniform(-2.0, 2.0, size=2) for _ in range(100)]  # exceeds the bound: exercises the clip
            log = run(env, events)
            np.savez(os.path.join(HERE, f"arm_env_{mode}_seed{seed}.npz"), np_seed=seed, **log)

    for mode in ("easy", "hard"):
        np.random.seed(7)
        env = arm_env.ArmEnv(mode=mode)
        run(env, [None, np.array([0.3, -0.4])])
        env.point_info[:] = env.arm_info[1, 2:4]             # truncating assignment into the int goal
        goal = env.point_info.copy()
        theta0 = env.arm_info[:, 1].copy()
        ends0 = env.arm_info[:, 2:4].copy()
        events = [np.zeros(2)] * 60 + [np.ones(2)] * 12 + [None] + [np.zeros(2)] * 3
        log = run(env, events)
        np.savez(os.path.join(HERE, f"arm_env_reach_{mode}.npz"), np_seed=7, goal0=goal, theta0=theta0,
                 ends0=ends0, **log)


def gen_memory(ddpg, cgan):
    rng = np.random.RandomState(5)
    cap = 64
    n = int(cap * 1.5)
    # 4-field layout
    np.random.seed(11)
    m4 = ddpg.Memory(cap, dims=2 * 7 + 2 + 1)
    rows4 = []
    for _ in range(n):
        s, a, r, s_ = rng.randn(7), rng.randn(2), float(rng.randn()), rng.randn(7)
        m4.store_transition(s, a, r, s_)
        rows4.append(np.hstack((s, a, [r], s_)))
    batch4 = m4.sample(16)
    # 6-field layout
    np.random.seed(12)
    m6 = cgan.Memory(cap, dims=3 * 7 + 2 + 2)
    rows6 = []
    for _ in range(n):
        s, s_, g, a, r, cr = rng.randn(7), rng.randn(7), rng.randn(7), rng.randn(2), float(rng.randn()), rng.randn(1)
        m6.store_transition(s, s_, g, a, r, cr)
        rows6.append(np.hstack((s, s_, g, a, [r], cr)))
    batch6 = m6.sample(16)
    np.savez(os.path.join(HERE, "memory_ring.npz"), capacity=cap,
             rows4=np.array(rows4), data4=m4.data, pointer4=m4.pointer, batch4=batch4, sample_seed4=11,
             rows6=np.array(rows6), data6=m6.data, pointer6=m6.pointer, batch6=batch6, sample_seed6=12)


def gen_sampler():
    out = {}
    for (N, n, seed) in [(5000, 16, 1), (5000, 32, 1), (5000, 4096, 1), (2 ** 20, 4096, 7), (1000003, 4096, 3)]:
        out[f"idx_{N}_{n}_{seed}"] = np.random.RandomState(seed).choice(N, size=n)
    # two consecutive draws from one stream (state continuity)
    rs = np.random.RandomState(9)
    out["idx_5000_4096_9_first"] = rs.choice(5000, size=4096)
    out["idx_5000_4096_9_second"] = rs.choice(5000, size=4096)
    np.savez(os.path.join(HERE, "sampler_indices.npz"), **out)


def main():
    _install_stand_ins()
    import arm_env
    import DDPG
    import Curious_GAN
    gen_arm_env(arm_env)
    gen_memory(DDPG, Curious_GAN)
    gen_sampler()
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
