#!/usr/bin/env python3
"""Rate of the reference's own single-transition loops (Curious_GAN.py:172-307) on the drop-in classes: every call
hands NumPy arrays over PCIe and waits for its result, nine network executions per transition once the replay is
full.  Beside it the same per-transition sequence on the NumPy restatement (host cores): that restatement is test
infrastructure, which is why this script lives with the tests.  pytest does not collect it; run it by name."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rl_gan_amd as R


def run(loop, batch, capacity=400, episodes=12):
    cfg = R.Settings(MEMORY_CAPACITY=capacity, BATCH_SIZE=batch, MAX_EPISODES=episodes)
    ag = R.Agent(cfg)
    loop(ag, episodes=2, log=None)                      # fills the replay (2 x 200 transitions), first calls
    torch.cuda.synchronize()
    p0, t0 = ag.M.pointer, time.perf_counter()
    loop(ag, episodes=episodes - 2, log=None)
    torch.cuda.synchronize()
    return (ag.M.pointer - p0) / (time.perf_counter() - t0)


def cpu(batch, n=300):
    from oracle.agent import CuriousAgent
    from oracle.arm_env import ScalarArmEnv as ArmEnv
    ag = CuriousAgent(seed=1234, dtype=np.float32)
    env = ArmEnv('hard')
    np.random.seed(1)
    s = env.reset()
    ring = np.random.uniform(-1, 1, (400, 25)).astype(np.float32)
    t0 = time.perf_counter()
    for i in range(n):
        a = np.clip(np.random.normal(ag.choose_action(s[None])[0], 1.0), -1, 1)
        g = ag.predict(s[None], a[None])[0]
        s_, r, done = env.step(a)
        ag.determine(s[None], a[None], g[None])
        ag.update(ring[np.random.choice(400, batch)])
        s = env.reset() if done or i % 200 == 199 else s_
    return n / (time.perf_counter() - t0)


if __name__ == "__main__":
    for b in (16, 32):
        print(f"batch {b:3d}: train (Curious_GAN + DDPG) {run(R.train, b):8.1f} transitions/s   train_DDPG {run(R.train_DDPG, b):8.1f} transitions/s"
              f"   NumPy restatement on the host {cpu(b):8.1f} transitions/s", flush=True)
