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
        # Per-step scalars the acting launch reads and advances live in two copies: a launch reads copy
        # `_cur` and writes the advanced values to the other one, so no workgroup has to wait for the rest
        # of the launch before it may overwrite them.
        self._var2 = torch.full((2, 1), cfg.var_init, dtype=torch.float64, device=dev)     # noise scale
        self._ctr2 = torch.zeros((2, 2), dtype=torch.int64, device=dev)                    # replay pointer, vector steps
        self._cur = 0
        self.u = torch.zeros((cfg.n_env, 2), dtype=torch.float64, device=dev)
        self.idx = torch.zeros(cfg.batch_size, dtype=torch.int64, device=dev)
        self.philox_seed = ((cfg.seed << 20) ^ (rank + 1)) or 1
        self.t = 0           # vector steps taken (host mirror of ctr[1])
        self.updates = 0
        self._graph = None   # after capture(): the two graphs of an iteration, indexed by `_cur`
        self._eval_pending = False
        self.reset_all()

    # ---- environment side ---------------------------------------------------------------------
    def reset_all(self) -> None:
        check(lib.rlgan_philox_uniform(self.u.data_ptr(), self.cfg.n_env, self.philox_seed, 2, 0, None, 0, _stream()),
              "rlgan_philox_uniform")
        self.env.reset(self.u)
        self.ep_t.zero_()

    @property
    def var(self) -> torch.Tensor:
        """Exploration noise scale (device float64 [1]), Curious_GAN.py:173,198."""
        return self._var2[self._cur]

    @property
    def ctr(self) -> torch.Tensor:
        """Device counters (int64 [2]): replay pointer, vector steps."""
        return self._ctr2[self._cur]

    def _act_args(self, noise: Optional[torch.Tensor] = None, cur: Optional[int] = None) -> _lib.Act:
        """The acting step of every environment with its bookkeeping: noise and goal draws (Philox, keyed
        by the device step counter), replay store, episode cut, counters, noise-scale decay.  Every
        per-step scalar is read from device memory, so the same launch can be replayed in a graph."""
        cfg, m = self.cfg, self.memory
        cur = self._cur if cur is None else cur
        return _lib.Act(cfg.s_dim, cfg.a_dim, cfg.n_env, self.actor.e.net(), self.gen.net.net(), self.disc.net.net(),
                        self.env.handle(), self.env.state32.data_ptr(), noise.data_ptr() if noise is not None else None,
                        self._var2[cur].data_ptr(), cfg.ita, m._data.data_ptr(), m.capacity, m.dims, m.ld, 0, 0,
                        self._ctr2[cur].data_ptr(), self.env.reward.data_ptr(), self.env.done.data_ptr(), self.philox_seed,
                        self.ep_t.data_ptr(), cfg.max_ep_steps, 1 if cfg.mode == 'hard' else 0, None,
                        self.n_resets.data_ptr(), cfg.var_decay, cfg.var_min,
                        self._ctr2[cur ^ 1].data_ptr(), self._var2[cur ^ 1].data_ptr())

    def _sample_args(self) -> _lib.Sample:
        m = self.memory
        return _lib.Sample(m.rng.state.data_ptr(), m.capacity, self.cfg.batch_size, self.idx.data_ptr())

    def _ring_fields(self):
        S, A = self.cfg.s_dim, self.cfg.a_dim
        d = self.memory._data
        return d[:, :S], d[:, S:2 * S], d[:, 2 * S:3 * S], d[:, 3 * S:3 * S + A]      # s, s_, g, a

    def rollout_step(self, noise: Optional[torch.Tensor] = None) -> None:
        """One acting step of every environment (Curious_GAN.py:190-198) and the episode cut; once the
        ring is full the same launch draws the indices of the next minibatch (Curious_GAN.py:199)."""
        full = self.memory.pointer + self.cfg.n_env > self.memory.capacity
        act = self._act_args(noise)
        sample = self._sample_args() if full else None
        check(lib.rlgan_rollout_eval_rows(act, None, sample, None, _stream()), "rlgan_rollout_eval_rows")
        self._cur ^= 1
        self.memory.pointer += self.cfg.n_env      # host mirror of ctr[0]
        self.t += 1                                # host mirror of ctr[1]

    # ---- learner side -------------------------------------------------------------------------
    def ready(self) -> bool:
        return self.memory.pointer > self.memory.capacity          # Curious_GAN.py:197

    def _update_body(self) -> None:
        """critic/D then actor/G steps on the minibatch ring[idx] (Curious_GAN.py:200-211)."""
        self.engine.update(self.actor, self.critic, self.gen, self.disc, self.memory._data, self.cfg.smooth,
                           with_eval=False, idx=self.idx)

    def _tail_body(self, cur: Optional[int] = None) -> None:
        """Everything between two updates that only reads the networks, as one launch: D_loss / G_loss of
        the update just done (Curious_GAN.py:212-213) as per-tile sums (`losses()` adds them up when
        asked: a total inside the launch would have one workgroup wait for all the others), the next
        acting step and the next index draw."""
        ev = self.engine.eval_args_after_update(self.gen, self.disc, self.cfg.batch_size, self.cfg.smooth, totals=False)
        act, sample = self._act_args(cur=cur), self._sample_args()
        self.engine._timed("rollout_eval_rows", lambda: check(
            lib.rlgan_rollout_eval_rows(act, ev, sample, None, _stream()), "rlgan_rollout_eval_rows"))

    def update(self, redraw: bool = True) -> None:
        """One minibatch update with its two losses (Curious_GAN.py:199-213), without acting: a fresh index draw
        (`M.sample`, Curious_GAN.py:199; redraw=False keeps the indices the last acting launch drew), the four steps,
        the two evaluations.  Losses: self.engine.losses."""
        if redraw:
            check(lib.rlgan_rollout_eval_rows(None, None, self._sample_args(), None, _stream()), "rlgan_rollout_eval_rows")
        self._update_body()
        s, s_, g, a = self._ring_fields()
        self.engine.gan_eval(self.gen, self.disc, g, s_, s, a, self.cfg.smooth, self.idx)
        self._eval_pending = False
        self.updates += 1

    def capture(self) -> None:
        """Record one full iteration (update, then losses + next acting step) as HIP graphs; `step()`
        then replays them.  Needs a full replay ring.  Two eager iterations run first so every work
        array exists."""
        assert self.ready(), "fill() the replay ring before capture()"
        assert self.engine.kernel_events is None
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(2):
                self.step()
        cur.wait_stream(s)
        torch.cuda.synchronize()
        graphs = [None, None]
        for cur in (self._cur, self._cur ^ 1):       # one graph per direction of the counter hand-over
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._update_body()
                self._tail_body(cur)
            graphs[cur] = g
        self._graph = graphs

    def step(self) -> None:
        """One iteration of the training loop.  While the ring fills: an acting step.  Afterwards: the
        update on the minibatch drawn after the previous acting step, then (one launch) its losses, the
        next acting step and the next draw -- the reference's act/store/sample/learn order, cut between
        `sample` and `learn` instead of before `act`."""
        if not self.ready():
            self.rollout_step()
            return
        if self._graph is not None:
            self._graph[self._cur].replay()
        else:
            self._update_body()
            self._tail_body()
        self._cur ^= 1
        self._eval_pending = True
        self.memory.pointer += self.cfg.n_env
        self.t += 1
        self.updates += 1

    def fill(self) -> None:
        """Roll out until the replay ring is full (no learning, like the reference's first
        MEMORY_CAPACITY transitions)."""
        while not self.ready():
            self.rollout_step()

    def losses(self) -> dict:
        """Losses of the last update as global-batch means.  Each rank holds its shard's sums divided
        by the global batch, so with more than one rank this is a collective: call it on every rank."""
        if self._eval_pending:
            self.engine.eval_totals(self.cfg.batch_size)
            self._eval_pending = False
        l = self.engine.losses.clone()
        self.exchange.all_reduce_sum(l)
        l = l.cpu().numpy()
        e = self.engine
        return {"critic": float(l[e.LOSS_CRITIC]), "D_step": float(l[e.LOSS_D_STEP]), "G_step": float(l[e.LOSS_G_STEP]),
                "D_loss": float(l[e.LOSS_D_EVAL]), "G_loss": float(l[e.LOSS_G_EVAL])}
