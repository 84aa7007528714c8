"""Vectorised Curious_GAN + DDPG trainer: thousands of arm environments in lock-step, one minibatch
update per vector step, data-parallel over the ranks of a torch.distributed job.

It reproduces the reference's loop (`train`, Curious_GAN.py:172-245) with the per-transition work
batched over `n_env` environments:

    a = actor(s); a = clip(normal(a, var), -1, 1); g = G(s, a); s', r, done = env.step(a)
    curious_r = r + ITA*(1 - D(s, a, g)); M.store(s, s', g, a, r, curious_r)
    if M.pointer > capacity: var = max(var*.9999, VAR_MIN); sample; critic, actor, D, G learn; D, G eval
    episode cut at MAX_EP_STEPS or done -> env.reset()

Per-rank state: its own environments, replay ring and index stream; replicated: all weights and
optimizer slots (kept identical by summing gradient buckets over ranks).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .arm_env import VecArmEnv
from .core import DeviceNet, _stream, device
from .engine import ActorState, AdamState, CriticState, GradExchange, StepEngine
from .replay import DeviceMemory, MT19937Device


@dataclass
class LearnerConfig:
    # names and defaults of the reference's constants (Curious_GAN.py:319-339)
    n_env: int = 4096
    batch_size: int = 4096          # BATCH_SIZE (reference default 16)
    memory_capacity: int = 1 << 20  # MEMORY_CAPACITY (reference default 5000)
    max_ep_steps: int = 200         # MAX_EP_STEPS
    lr_a: float = 1e-4
    lr_c: float = 1e-4
    lr_d: float = 1e-4
    lr_g: float = 1e-4
    gamma: float = 0.9
    replace_iter_a: int = 1100
    replace_iter_c: int = 1000
    var_init: float = 2.0           # Curious_GAN.py:173
    var_min: float = 0.1            # VAR_MIN
    var_decay: float = 0.9999       # Curious_GAN.py:198
    ita: float = 0.01               # ITA
    smooth: float = 0.0
    mode: str = 'hard'
    seed: int = 1
    critic_trains_actor: bool = True
    s_dim: int = 7
    a_dim: int = 2


class CuriousLearner:
    def __init__(self, cfg: Optional[LearnerConfig] = None, exchange: Optional[GradExchange] = None):
        self.cfg = cfg = cfg or LearnerConfig()
        self.exchange = exchange or GradExchange()
        rank = self.exchange.rank
        dev = device()
        S, A = cfg.s_dim, cfg.a_dim
        # identical weights on every rank: one seeded stream, construction order of the reference
        # graph (Curious_GAN.py:355-360): actor eval, actor target, critic eval, critic target, G, D
        rng = np.random.RandomState(cfg.seed)
        mk = lambda kind, train: DeviceNet(kind, S, A, trainable=train, rng=rng)
        self.actor = ActorState(e=mk(_lib.NET_ACTOR, True), t=mk(_lib.NET_ACTOR, False), lr=cfg.lr_a, period=cfg.replace_iter_a)
        self.critic = CriticState(e=mk(_lib.NET_CRITIC, True), t=mk(_lib.NET_CRITIC, False), lr=cfg.lr_c, gamma=cfg.gamma,
                                  period=cfg.replace_iter_c, trains_actor=cfg.critic_trains_actor)
        self.gen = AdamState(net=mk(_lib.NET_GENERATOR, True), lr=cfg.lr_g)
        self.disc = AdamState(net=mk(_lib.NET_DISCRIMINATOR, True), lr=cfg.lr_d)
        self.engine = StepEngine(S, A, self.exchange)
        # per-rank streams: environments, exploration noise, replay indices
        self.env = VecArmEnv(cfg.n_env, cfg.mode)
        self.ep_t = torch.zeros(cfg.n_env, dtype=torch.int32, device=dev)
        self.n_resets = torch.zeros(1, dtype=torch.int32, device=dev)
        self.memory = DeviceMemory(cfg.memory_capacity, 3 * S + A + 2, rng=MT19937Device(cfg.seed + 7919 * rank))
        self.var = torch.full((1,), cfg.var_init, dtype=torch.float32, device=dev)
        self.noise = torch.zeros((cfg.n_env, A), dtype=torch.float32, device=dev)
        self.u = torch.zeros((cfg.n_env, 2), dtype=torch.float64, device=dev)
        self.idx = torch.zeros(cfg.batch_size, dtype=torch.int64, device=dev)
        self.batch = torch.zeros((cfg.batch_size, 3 * S + A + 2), dtype=torch.float32, device=dev)
        self.philox_seed = (cfg.seed << 20) ^ (rank + 1)
        self.ctr = torch.zeros(2, dtype=torch.int64, device=dev)   # device counters: replay pointer, vector steps
        self.t = 0           # vector steps taken (host mirror of ctr[1])
        self.updates = 0
        self._graph = None
        self.reset_all()

    # ---- environment side ---------------------------------------------------------------------
    def reset_all(self) -> None:
        check(lib.rlgan_philox_uniform(self.u.data_ptr(), self.cfg.n_env, self.philox_seed, 2, 0, None, 0, _stream()),
              "rlgan_philox_uniform")
        self.env.reset(self.u)
        self.ep_t.zero_()

    def _rollout_body(self, noise: Optional[torch.Tensor] = None) -> None:
        """Acting step + episode cut.  Every per-step scalar (replay position, random-stream offset)
        is read from the device counters `self.ctr`, so the same launches can be replayed as a graph."""
        cfg = self.cfg
        n = cfg.n_env
        step = self.ctr[1:].data_ptr()
        if noise is None:
            check(lib.rlgan_philox_normal(self.noise.data_ptr(), n, self.philox_seed, 0, 0, step, n, _stream()),
                  "rlgan_philox_normal")
            noise = self.noise
        m = self.memory
        check(lib.rlgan_act_rows(cfg.s_dim, cfg.a_dim, self.actor.e.net(), self.gen.net.net(), self.disc.net.net(),
                                 self.env.handle(), n, self.env.state32.data_ptr(), noise.data_ptr(), self.var.data_ptr(),
                                 cfg.ita, m._data.data_ptr(), m.capacity, m.dims, m.ld, 0, self.ctr.data_ptr(),
                                 self.env.reward.data_ptr(), self.env.done.data_ptr(), _stream()), "rlgan_act_rows")
        check(lib.rlgan_philox_uniform(self.u.data_ptr(), n, self.philox_seed, 1, n, step, n, _stream()), "rlgan_philox_uniform")
        check(lib.rlgan_env_autoreset(self.env.handle(), n, 1 if cfg.mode == 'hard' else 0, self.env.done.data_ptr(),
                                      self.ep_t.data_ptr(), cfg.max_ep_steps, self.u.data_ptr(),
                                      self.env.state32.data_ptr(), self.n_resets.data_ptr(), self.ctr.data_ptr(), _stream()),
              "rlgan_env_autoreset")

    def rollout_step(self, noise: Optional[torch.Tensor] = None) -> None:
        """One acting step of every environment (Curious_GAN.py:190-195) and the episode cut."""
        self._rollout_body(noise)
        self.memory.pointer += self.cfg.n_env      # host mirror of ctr[0]
        self.t += 1                                # host mirror of ctr[1]

    # ---- learner side -------------------------------------------------------------------------
    def ready(self) -> bool:
        return self.memory.pointer > self.memory.capacity          # Curious_GAN.py:197

    def _update_body(self) -> None:
        cfg = self.cfg
        self.var.mul_(cfg.var_decay).clamp_(min=cfg.var_min)        # Curious_GAN.py:198
        self.memory.sample_indices(cfg.batch_size, self.idx)        # Curious_GAN.py:199
        self.memory.gather(self.idx, self.batch)
        self.engine.update(self.actor, self.critic, self.gen, self.disc, self.batch, cfg.smooth)

    def update(self) -> None:
        """One minibatch update (Curious_GAN.py:198-213).  Losses: self.engine.losses."""
        self._update_body()
        self.updates += 1

    def capture(self) -> None:
        """Record one full iteration (acting step + update) as a HIP graph; `step()` then replays it.
        Needs a full replay ring.  Two eager iterations run first so every work array exists."""
        assert self.ready(), "fill() the replay ring before capture()"
        assert self.engine.kernel_events is None
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(2):
                self.rollout_step()
                self.update()
        cur.wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._rollout_body()
            self._update_body()
        self._graph = g

    def step(self) -> None:
        """One iteration of the training loop: act, store, learn when the ring is full."""
        if self._graph is not None:
            self._graph.replay()
            self.memory.pointer += self.cfg.n_env
            self.t += 1
            self.updates += 1
            return
        self.rollout_step()
        if self.ready():
            self.update()

    def fill(self) -> None:
        """Roll out until the replay ring is full (no learning, like the reference's first
        MEMORY_CAPACITY transitions)."""
        while not self.ready():
            self.rollout_step()

    def losses(self) -> dict:
        """Losses of the last update as global-batch means.  Each rank holds its shard's sums divided
        by the global batch, so with more than one rank this is a collective: call it on every rank."""
        l = self.engine.losses.clone()
        self.exchange.all_reduce_sum(l)
        l = l.cpu().numpy()
        e = self.engine
        return {"critic": float(l[e.LOSS_CRITIC]), "D_step": float(l[e.LOSS_D_STEP]), "G_step": float(l[e.LOSS_G_STEP]),
                "D_loss": float(l[e.LOSS_D_EVAL]), "G_loss": float(l[e.LOSS_G_EVAL])}
