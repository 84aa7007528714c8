"""CPU restatement of the vectorised training loop: n environments in lock-step, one update per
vector step.  TEST INFRASTRUCTURE ONLY.

Per transition it is the reference's loop body (Curious_GAN.py:186-234):
    a = clip(normal(actor(s), var), -1, 1); g = G(s, a); s', r, done = env.step(a)
    curious_r = r + ITA*(1 - D(s, a, g)); M.store(s, s', g, a, r, curious_r)
    if M.pointer > capacity: var = max(var*.9999, VAR_MIN); sample; critic, actor, D, G learn; D, G eval
    episode cut at MAX_EP_STEPS or done -> env.reset()
with the choices a batched loop has to make, stated here once: all environments act on the same
weights; noise and goal placement come from a counter-based Philox stream (index = step*n + env),
minibatch indices from a private MT19937 stream (np.random.choice semantics); the state handed to the
networks and stored in the ring is float32.
"""
from __future__ import annotations

import numpy as np

from .agent import CuriousAgent
from .arm_env import VecArmEnv
from .mt19937 import MT19937
from . import philox

STREAM_NOISE, STREAM_GOAL, STREAM_FIRST_GOAL = 0, 1, 2


class VecLoop:
    def __init__(self, agent: CuriousAgent, n_env: int, batch: int, capacity: int, mode: str, philox_seed: int,
                 mt_seed: int, max_ep_steps: int = 200, var_init: float = 2.0, var_min: float = 0.1, var_decay: float = 0.9999):
        self.ag, self.n, self.batch, self.cap = agent, n_env, batch, capacity
        self.seed, self.max_ep = philox_seed, max_ep_steps
        self.var, self.var_min, self.var_decay = var_init, var_min, var_decay
        self.env = VecArmEnv(n_env, mode)
        u = np.array([philox.uniform_pair(i, STREAM_FIRST_GOAL, philox_seed) for i in range(n_env)])
        self.state32 = self.env.reset(np.ones(n_env, bool), u).astype(np.float32)
        self.ep_t = np.zeros(n_env, np.int64)
        self.ring = np.zeros((capacity, 25), np.float32)
        self.pointer, self.step_count, self.n_resets = 0, 0, 0
        self.mt = MT19937(mt_seed)
        self.idx = None

    def act(self):
        ag, n = self.ag, self.n
        s = self.state32
        z = np.array([philox.normal_pair(self.step_count * n + i, STREAM_NOISE, self.seed) for i in range(n)], np.float32)
        a32 = ag.choose_action(s.astype(np.float64)).astype(np.float32)
        a64 = np.clip(a32.astype(np.float64) + self.var * z.astype(np.float64), -1.0, 1.0)   # np.random.normal(a, var) in float64
        a = a64.astype(np.float32)
        g = ag.predict(s.astype(np.float64), a.astype(np.float64)).astype(np.float32)
        s_, r, done = self.env.step(a64)
        d = ag.determine(s.astype(np.float64), a.astype(np.float64), g.astype(np.float64)).ravel()
        cr = r.astype(np.float32) + np.float32(ag.hp.ita) * d.astype(np.float32)
        rows = np.hstack([s, s_.astype(np.float32), g, a, r.astype(np.float32)[:, None], cr[:, None]])
        self.ring[(self.pointer + np.arange(n)) % self.cap] = rows
        # episode cut (Curious_GAN.py:180,222-234)
        cut = done | (self.ep_t >= self.max_ep - 1)
        if cut.any():
            u = np.zeros((n, 2))
            for i in np.nonzero(cut)[0]:
                u[i] = philox.uniform_pair(n + self.step_count * n + int(i), STREAM_GOAL, self.seed)
            s_ = np.where(cut[:, None], self.env.reset(cut, u), s_)
            self.n_resets += int(cut.sum())
        self.ep_t = np.where(cut, 0, self.ep_t + 1)
        self.state32 = s_.astype(np.float32)
        self.pointer += n
        self.step_count += 1
        if self.pointer > self.cap:
            self.var = max(self.var * self.var_decay, self.var_min)
            self.idx = self.mt.choice(self.cap, self.batch)

    def update(self):
        return self.ag.update(self.ring[self.idx].astype(np.float64))
