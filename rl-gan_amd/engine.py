"""The update steps of the reference, expressed as sequences of librlgan_amd kernels.

Sequence per minibatch (Curious_GAN.py:208-213):
    (1) critic.learn  (2) actor.learn  (3) discriminator.learn  (4) generator.learn
    (5) discriminator.eval  (6) generator.eval
(1)->(2) and (3)->(4) are two independent chains, so the fused `update` runs them as two phases,
{critic, D} then {actor, G}: one launch per stage of a phase (row passes of both steps tile by
tile, weight gradients, chunk reduction, optimizers) and ONE gradient exchange over a flat bucket.

Data-parallel contract (no counterpart in the reference, which is single-process): every rank holds
identical weights, computes gradients of its own shard with mean-type losses already divided by the
GLOBAL batch, the buckets are summed with all_reduce, and every rank applies the same update.  The
actor's policy gradient is a batch SUM in the reference (DDPG.py:109), so it is summed unscaled.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .core import (NULL_ACTS, NULL_NET, DeviceNet, Workspace, _stream, adam_job, device, launch_apply, launch_reduce,
                   launch_wgrads, mat, ptr, reduce_job, rmsprop_job, rows16, sum_parts, wgrad_job)


class GradExchange:
    """Sum of flat gradient buckets over the data-parallel ranks (RCCL through torch.distributed on
    GPUs; any initialised backend works, which is how the CPU tests drive it with gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        up = dist.is_available() and dist.is_initialized()
        # RLGAN_FORCE_EXCHANGE=1 keeps the collective in the step on a one-rank job (rehearsal of the
        # multi-GPU launch sequence, graph capture included, on a single GPU)
        self.enabled = up and (dist.get_world_size(group) > 1 or os.environ.get("RLGAN_FORCE_EXCHANGE") == "1")
        self.group = group
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

    def new_bucket(self, n: int) -> torch.Tensor:
        """A zeroed flat fp32 bucket of n floats for `all_reduce_sum`."""
        return torch.zeros(n, dtype=torch.float32, device=device())

    def all_reduce_sum(self, bucket: torch.Tensor) -> None:
        if not self.enabled:
            return
        if bucket.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # test rigs only (several ranks on one GPU): gloo moves the bucket through host memory
            host = bucket.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
            bucket.copy_(host)
            return
        self.dist.all_reduce(bucket, op=self.dist.ReduceOp.SUM, group=self.group)


@dataclass
class ActorState:
    e: DeviceNet
    t: DeviceNet
    lr: float
    period: int
    ms: torch.Tensor = None            # A_train RMSProp slot (DDPG.py:112)
    counter: torch.Tensor = None       # t_replace_counter (DDPG.py:64), device int32

    def __post_init__(self):
        dev = device()
        self.ms = torch.ones(self.e.n_params, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)


@dataclass
class CriticState:
    e: DeviceNet
    t: DeviceNet
    lr: float
    gamma: float
    period: int
    trains_actor: bool = True          # minimize() without var_list, DDPG.py:148
    ms: torch.Tensor = None            # C_train slots for the critic variables
    ms_actor: torch.Tensor = None      # C_train slots for the actor variables
    counter: torch.Tensor = None

    def __post_init__(self):
        dev = device()
        self.ms = torch.ones(self.e.n_params, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)

    def ensure_actor_slots(self, n: int):
        if self.ms_actor is None:
            self.ms_actor = torch.ones(n, dtype=torch.float32, device=device())


@dataclass
class AdamState:
    net: DeviceNet
    lr: float
    m: torch.Tensor = None
    v: torch.Tensor = None
    step: torch.Tensor = None          # device int32, t of the Adam bias correction

    def __post_init__(self):
        dev = device()
        self.m = torch.zeros(self.net.n_params, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.net.n_params, dtype=torch.float32, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)


class StepEngine:
    """Owns the work arrays (per batch size) and runs the steps.  Stateless w.r.t. the networks."""

    LOSS_CRITIC, LOSS_D_STEP, LOSS_G_STEP, LOSS_D_EVAL, LOSS_G_EVAL = range(5)

    def __init__(self, s_dim: int, a_dim: int, exchange: Optional[GradExchange] = None, chains: Optional[int] = None):
        self.s_dim, self.a_dim = s_dim, a_dim
        self.exchange = exchange or GradExchange()
        self._ws: Dict[Tuple, object] = {}
        self.losses = torch.zeros(8, dtype=torch.float32, device=device())
        self.kernel_events = None     # set to {} to record (start, end) event pairs per kernel launch
        # Form of a multi-rank update.  chains = 1 (default): the two fused phase launches {critic, D} and {actor, G}
        # with one gradient exchange each.  chains = 2 (argument, or RLGAN_CHAINS=2): critic step -> actor step and
        # D step -> G step as two chains on two streams, each step with its own exchange, so that one chain computes
        # while the other waits for its all-reduce.  Both give the same bits; on one rank there is nothing to wait
        # for and the fused phases are always used unless the exchange is forced on.
        self.set_chains(chains if chains is not None else int(os.environ.get("RLGAN_CHAINS", "1")))

    def set_chains(self, chains: int) -> None:
        """Select the form of the multi-rank update (1 = fused phases, 2 = two chains); graphs captured before the
        call keep the form they were captured with."""
        if chains not in (1, 2):
            raise ValueError("chains must be 1 or 2")
        self.two_chains = chains == 2

    def _timed(self, name: str, launch):
        """Run `launch()`; when kernel_events is a dict, bracket it with events on the launch stream."""
        if self.kernel_events is None:
            return launch()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = launch()
        b.record()
        self.kernel_events.setdefault(name, []).append((a, b))
        return out

    def _side_stream(self) -> "torch.cuda.Stream":
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def kernel_times_ms(self) -> Dict[str, list]:
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in (self.kernel_events or {}).items()}

    # ---- work arrays ------------------------------------------------------------------------
    def _get(self, key, make):
        if key not in self._ws:
            self._ws[key] = make()
        return self._ws[key]

    def workspace(self, tag: str, net: DeviceNet, rows: int) -> Workspace:
        return self._get((tag, net.kind, rows), lambda: Workspace(net, rows))

    def loss_parts(self, tag: str, n: int) -> torch.Tensor:
        return self._get(("lp", tag, n), lambda: torch.zeros(n, dtype=torch.float32, device=device()))

    def bucket(self, tag: str, n: int) -> torch.Tensor:
        return self._get(("bucket", tag, n), lambda: self.exchange.new_bucket(n))

    def _inv_gb(self, n_rows: int) -> float:
        return 1.0 / float(n_rows * self.exchange.world)

    # ---- argument records of the four row passes ---------------------------------------------------
    @staticmethod
    def _rows(s, idx) -> int:
        return s.shape[0] if idx is None else idx.numel()

    def _critic_args(self, C: CriticState, A: ActorState, s, a, r, s_, idx=None):
        n = self._rows(s, idx)
        R = rows16(n)
        wc = self.workspace("c", C.e, R)
        wa = self.workspace("ca", A.e, R) if C.trains_actor else None
        lp = self.loss_parts("c", R // 16)
        args = _lib.CriticStep(self.s_dim, self.a_dim, n, C.e.net(True), C.t.net(), A.t.net(),
                               A.e.net(True) if C.trains_actor else NULL_NET, mat(s), mat(a), mat(r), mat(s_),
                               C.gamma, self._inv_gb(n), wc.acts(), wa.acts() if wa is not None else NULL_ACTS, lp.data_ptr(),
                               ptr(idx), C.counter.data_ptr())
        return args, wc, wa, lp

    def _actor_args(self, A: ActorState, C: CriticState, s, idx=None):
        n = self._rows(s, idx)
        wa = self.workspace("a", A.e, rows16(n))
        return _lib.ActorStep(self.s_dim, self.a_dim, n, A.e.net(True), C.e.net(True), mat(s), wa.acts(), ptr(idx),
                              A.counter.data_ptr()), wa

    def _disc_args(self, D: AdamState, g_data, s_, s, a, smooth, idx=None):
        n = self._rows(s, idx)
        R = rows16(n)
        wd = self.workspace("d", D.net, 2 * R)
        lp = self.loss_parts("d", R // 16)
        args = _lib.DiscStep(self.s_dim, self.a_dim, n, D.net.net(True), mat(s), mat(g_data), mat(a), mat(s_),
                             1.0 - smooth, self._inv_gb(n), wd.acts(), lp.data_ptr(), ptr(idx), D.step.data_ptr())
        return args, wd, lp

    def _gen_args(self, G: AdamState, D: AdamState, s, a, idx=None):
        n = self._rows(s, idx)
        R = rows16(n)
        wg = self.workspace("g", G.net, R)
        lp = self.loss_parts("g", R // 16)
        args = _lib.GenStep(self.s_dim, self.a_dim, n, G.net.net(True), D.net.net(True), mat(s), mat(a),
                            self._inv_gb(n), wg.acts(), lp.data_ptr(), ptr(idx), G.step.data_ptr())
        return args, wg, lp

    # ---- job records ------------------------------------------------------------------------------
    def _critic_jobs(self, C, A, wc, wa, lp, n):
        wj = [wgrad_job(C.e, wc)]
        rj = [reduce_job(C.e, wc, lp, lp.numel(), self._inv_gb(n), self.losses[self.LOSS_CRITIC:])]
        if wa is not None:
            wj.append(wgrad_job(A.e, wa))
            rj.append(reduce_job(A.e, wa))
        return wj, rj

    def _critic_apply_jobs(self, C, A, wc, wa):
        # RMSProp on the critic, hard target copy on the period (DDPG.py:176-178); the actor eval net
        # moves too when the critic's train-op reaches it (DDPG.py:148), with slots of its own
        jobs = [rmsprop_job(C.e, wc.grad, C.ms, C.lr, target=C.t, tau=1.0, counter=C.counter, period=C.period)]
        if wa is not None:
            C.ensure_actor_slots(A.e.n_params)
            jobs.append(rmsprop_job(A.e, wa.grad, C.ms_actor, C.lr))
        return jobs

    def _actor_apply_job(self, A, wa):
        # RMSPropOptimizer(-lr): ascent (DDPG.py:112); hard target copy on the period (DDPG.py:99-101)
        return rmsprop_job(A.e, wa.grad, A.ms, -A.lr, target=A.t, tau=1.0, counter=A.counter, period=A.period)

    # ---- (1) critic.learn ---------------------------------------------------------------------
    def critic_rows(self, C: CriticState, A: ActorState, s, a, r, s_):
        args, wc, wa, lp = self._critic_args(C, A, s, a, r, s_)
        self._timed("critic_step_rows", lambda: check(lib.rlgan_critic_step_rows(args, _stream()), "rlgan_critic_step_rows"))
        return wc, wa, lp

    def critic_step(self, C, A, s, a, r, s_):
        wc, wa, lp = self.critic_rows(C, A, s, a, r, s_)
        wj, rj = self._critic_jobs(C, A, wc, wa, lp, s.shape[0])
        launch_wgrads(wj, self.s_dim, self.a_dim)
        launch_reduce(rj)
        self.exchange.all_reduce_sum(wc.grad)
        if wa is not None:
            self.exchange.all_reduce_sum(wa.grad)
        launch_apply(self._critic_apply_jobs(C, A, wc, wa))
        return wc, wa

    # ---- (2) actor.learn ----------------------------------------------------------------------
    def actor_rows(self, A: ActorState, C: CriticState, s):
        args, wa = self._actor_args(A, C, s)
        self._timed("actor_step_rows", lambda: check(lib.rlgan_actor_step_rows(args, _stream()), "rlgan_actor_step_rows"))
        return wa

    def actor_step(self, A, C, s):
        wa = self.actor_rows(A, C, s)
        launch_wgrads([wgrad_job(A.e, wa)], self.s_dim, self.a_dim)
        launch_reduce([reduce_job(A.e, wa)])
        self.exchange.all_reduce_sum(wa.grad)
        launch_apply([self._actor_apply_job(A, wa)])
        return wa

    # ---- (3) discriminator.learn --------------------------------------------------------------
    def disc_rows(self, D: AdamState, g_data, s_, s, a, smooth: float = 0.0):
        args, wd, lp = self._disc_args(D, g_data, s_, s, a, smooth)
        self._timed("disc_step_rows", lambda: check(lib.rlgan_disc_step_rows(args, _stream()), "rlgan_disc_step_rows"))
        return wd, lp

    def disc_step(self, D, g_data, s_, s, a, smooth: float = 0.0):
        wd, lp = self.disc_rows(D, g_data, s_, s, a, smooth)
        launch_wgrads([wgrad_job(D.net, wd)], self.s_dim, self.a_dim)
        launch_reduce([reduce_job(D.net, wd, lp, lp.numel(), self._inv_gb(s.shape[0]), self.losses[self.LOSS_D_STEP:])])
        self.exchange.all_reduce_sum(wd.grad)
        launch_apply([adam_job(D.net, wd.grad, D.m, D.v, D.step, D.lr)])
        return wd

    # ---- (4) generator.learn ------------------------------------------------------------------
    def gen_rows(self, G: AdamState, D: AdamState, s, a):
        args, wg, lp = self._gen_args(G, D, s, a)
        self._timed("gen_step_rows", lambda: check(lib.rlgan_gen_step_rows(args, _stream()), "rlgan_gen_step_rows"))
        return wg, lp

    def gen_step(self, G, D, s, a):
        wg, lp = self.gen_rows(G, D, s, a)
        launch_wgrads([wgrad_job(G.net, wg)], self.s_dim, self.a_dim)
        launch_reduce([reduce_job(G.net, wg, lp, lp.numel(), self._inv_gb(s.shape[0]), self.losses[self.LOSS_G_STEP:])])
        self.exchange.all_reduce_sum(wg.grad)
        launch_apply([adam_job(G.net, wg.grad, G.m, G.v, G.step, G.lr)])
        return wg

    # ---- (5)+(6) post-update losses -------------------------------------------------------------
    def ticket(self) -> torch.Tensor:
        """The end-of-launch ticket (one device int32, zero between launches) shared by this engine's calls."""
        return self._get(("ticket",), lambda: torch.zeros(1, dtype=torch.int32, device=device()))

    def eval_args(self, G: AdamState, D: AdamState, g_data, s_, s, a, smooth: float = 0.0, idx=None) -> _lib.Eval:
        n = self._rows(s, idx)
        lp = self.loss_parts("e", 2 * (rows16(n) // 16))
        out = self.losses[self.LOSS_D_EVAL:self.LOSS_D_EVAL + 2]
        return _lib.Eval(self.s_dim, self.a_dim, n, G.net.net(), D.net.net(), mat(s), mat(g_data), mat(a), mat(s_),
                         1.0 - smooth, self._inv_gb(n), lp.data_ptr(), out.data_ptr(), ptr(idx), None, None)

    def eval_args_after_update(self, G: AdamState, D: AdamState, n: int, smooth: float = 0.0,
                               totals: bool = True) -> _lib.Eval:
        """Evaluation of the minibatch the last `update` of `n` rows ran on, read from the input tiles
        its D and G steps left in the work arrays (no gather, the replay ring is not touched).
        With totals=False the launch only leaves the per-tile sums; `eval_totals(n)` adds them up later."""
        R = rows16(n)
        lp = self.loss_parts("e", 2 * (R // 16))
        out = self.losses[self.LOSS_D_EVAL:self.LOSS_D_EVAL + 2] if totals else None
        wd, wg = self.workspace("d", D.net, 2 * R), self.workspace("g", G.net, R)
        none = mat(None)
        return _lib.Eval(self.s_dim, self.a_dim, n, G.net.net(), D.net.net(), none, none, none, none,
                         1.0 - smooth, self._inv_gb(n), lp.data_ptr(), ptr(out), None,
                         wd.x0.data_ptr(), wg.x0.data_ptr())

    def eval_totals(self, n: int) -> None:
        """D_loss and G_loss from the per-tile sums an evaluation of `n` rows left (same order of
        additions as the in-launch totals)."""
        nt = rows16(n) // 16
        sum_parts(self.loss_parts("e", 2 * nt), nt, 2, self._inv_gb(n), self.losses[self.LOSS_D_EVAL:self.LOSS_D_EVAL + 2])

    def gan_eval(self, G: AdamState, D: AdamState, g_data, s_, s, a, smooth: float = 0.0, idx=None):
        """D_loss and G_loss of this rank's shard, divided by the global batch (summing over ranks gives
        the global means); returns the device view [2] of self.losses that receives them."""
        args = self.eval_args(G, D, g_data, s_, s, a, smooth, idx)
        self._timed("gan_eval_rows", lambda: check(lib.rlgan_gan_eval_rows(args, self.ticket().data_ptr(), _stream()),
                                                   "rlgan_gan_eval_rows"))
        return self.losses[self.LOSS_D_EVAL:self.LOSS_D_EVAL + 2]

    # ---- fused update: two phases, one launch per stage, one exchange per phase ---------------------
    def update(self, A: ActorState, C: CriticState, G: AdamState, D: AdamState, batch: torch.Tensor,
               smooth: float = 0.0, with_eval: bool = True, use_curious_reward: bool = True,
               idx: Optional[torch.Tensor] = None):
        """One reference update on a [B, 3S+A+2] fp32 batch in the 6-field replay layout
        (Curious_GAN.py:162,199-213).  With `idx` (int64 [B], device) `batch` is the replay ring itself and
        minibatch row i is ring row idx[i].  Losses land in self.losses (device)."""
        S, Ad = self.s_dim, self.a_dim
        n = self._rows(batch, idx)
        b_s, b_s_, b_g = batch[:, :S], batch[:, S:2 * S], batch[:, 2 * S:3 * S]
        b_a = batch[:, 3 * S:3 * S + Ad]
        b_r = batch[:, 3 * S + Ad + (1 if use_curious_reward else 0)]
        nc, na, nd, ng = C.e.n_params, A.e.n_params, D.net.n_params, G.net.n_params
        if self.two_chains and self.exchange.enabled:
            self._update_two_chains(A, C, G, D, b_s, b_a, b_r, b_s_, b_g, smooth, idx, n)
            if with_eval:
                self.gan_eval(G, D, b_g, b_s_, b_s, b_a, smooth, idx)
            return
        # phase 1: critic step and discriminator step, tile by tile in one launch
        cargs, wc, wa, lpc = self._critic_args(C, A, b_s, b_a, b_r, b_s_, idx)
        dargs, wd, lpd = self._disc_args(D, b_g, b_s_, b_s, b_a, smooth, idx)
        b1 = self.bucket("p1", nc + na + nd)
        wc.grad = b1[:nc]
        if wa is not None:
            wa.grad = b1[nc:nc + na]
        wd.grad = b1[nc + na:]
        self._timed("phase1_rows", lambda: check(lib.rlgan_phase1_rows(cargs, dargs, _stream()), "rlgan_phase1_rows"))
        wj, rj = self._critic_jobs(C, A, wc, wa, lpc, n)
        launch_wgrads(wj + [wgrad_job(D.net, wd)], S, Ad)
        self._reduce_exchange_apply(
            rj + [reduce_job(D.net, wd, lpd, lpd.numel(), self._inv_gb(n), self.losses[self.LOSS_D_STEP:])], b1,
            self._critic_apply_jobs(C, A, wc, wa) + [adam_job(D.net, wd.grad, D.m, D.v, D.step, D.lr)])
        # phase 2: actor step and generator step (both read the nets phase 1 just updated)
        aargs, wa2 = self._actor_args(A, C, b_s, idx)
        gargs, wg, lpg = self._gen_args(G, D, b_s, b_a, idx)
        b2 = self.bucket("p2", na + ng)
        wa2.grad = b2[:na]
        wg.grad = b2[na:]
        self._timed("phase2_rows", lambda: check(lib.rlgan_phase2_rows(aargs, gargs, _stream()), "rlgan_phase2_rows"))
        launch_wgrads([wgrad_job(A.e, wa2), wgrad_job(G.net, wg)], S, Ad)
        self._reduce_exchange_apply(
            [reduce_job(A.e, wa2), reduce_job(G.net, wg, lpg, lpg.numel(), self._inv_gb(n), self.losses[self.LOSS_G_STEP:])], b2,
            [self._actor_apply_job(A, wa2), adam_job(G.net, wg.grad, G.m, G.v, G.step, G.lr)])
        if with_eval:
            self.gan_eval(G, D, b_g, b_s_, b_s, b_a, smooth, idx)

    def _update_two_chains(self, A, C, G, D, b_s, b_a, b_r, b_s_, b_g, smooth, idx, n) -> None:
        """The same four steps as two chains on two streams (the DDPG nets and the GAN nets share nothing inside an
        update).  The steps are issued in the order their gradients become ready -- D, critic, G, actor -- because the
        collectives of one process group run one after the other in issue order."""
        S, Ad = self.s_dim, self.a_dim
        nc, na, nd, ng = C.e.n_params, A.e.n_params, D.net.n_params, G.net.n_params
        cur = torch.cuda.current_stream()
        side = self._get(("side_stream",), lambda: torch.cuda.Stream())
        side.wait_stream(cur)
        with torch.cuda.stream(side):                            # D step
            dargs, wd, lpd = self._disc_args(D, b_g, b_s_, b_s, b_a, smooth, idx)
            wd.grad = self.bucket("c2d", nd)
            self._timed("disc_step_rows", lambda: check(lib.rlgan_disc_step_rows(dargs, _stream()), "rlgan_disc_step_rows"))
            launch_wgrads([wgrad_job(D.net, wd)], S, Ad)
            self._reduce_exchange_apply([reduce_job(D.net, wd, lpd, lpd.numel(), self._inv_gb(n), self.losses[self.LOSS_D_STEP:])],
                                        wd.grad, [adam_job(D.net, wd.grad, D.m, D.v, D.step, D.lr)])
        cargs, wc, wa, lpc = self._critic_args(C, A, b_s, b_a, b_r, b_s_, idx)    # critic step
        bc = self.bucket("c2c", nc + (na if wa is not None else 0))
        wc.grad = bc[:nc]
        if wa is not None:
            wa.grad = bc[nc:]
        self._timed("critic_step_rows", lambda: check(lib.rlgan_critic_step_rows(cargs, _stream()), "rlgan_critic_step_rows"))
        wj, rj = self._critic_jobs(C, A, wc, wa, lpc, n)
        launch_wgrads(wj, S, Ad)
        self._reduce_exchange_apply(rj, bc, self._critic_apply_jobs(C, A, wc, wa))
        with torch.cuda.stream(side):                            # G step (reads the D the side stream just updated)
            gargs, wg, lpg = self._gen_args(G, D, b_s, b_a, idx)
            wg.grad = self.bucket("c2g", ng)
            self._timed("gen_step_rows", lambda: check(lib.rlgan_gen_step_rows(gargs, _stream()), "rlgan_gen_step_rows"))
            launch_wgrads([wgrad_job(G.net, wg)], S, Ad)
            self._reduce_exchange_apply([reduce_job(G.net, wg, lpg, lpg.numel(), self._inv_gb(n), self.losses[self.LOSS_G_STEP:])],
                                        wg.grad, [adam_job(G.net, wg.grad, G.m, G.v, G.step, G.lr)])
        aargs, wa2 = self._actor_args(A, C, b_s, idx)            # actor step (reads the critic and actor just updated)
        wa2.grad = self.bucket("c2a", na)
        self._timed("actor_step_rows", lambda: check(lib.rlgan_actor_step_rows(aargs, _stream()), "rlgan_actor_step_rows"))
        launch_wgrads([wgrad_job(A.e, wa2)], S, Ad)
        self._reduce_exchange_apply([reduce_job(A.e, wa2)], wa2.grad, [self._actor_apply_job(A, wa2)])
        cur.wait_stream(side)

    def _reduce_exchange_apply(self, reduce_jobs, bucket: torch.Tensor, apply_jobs) -> None:
        """Chunk reduction -> sum over ranks -> optimizers.  With one rank nothing happens in between, so
        the reduction runs inside the optimizer launch."""
        if self.exchange.enabled:
            launch_reduce(reduce_jobs)
            self._timed(f"exchange_{bucket.numel() * 4 // 1024}KiB", lambda: self.exchange.all_reduce_sum(bucket))
            launch_apply(apply_jobs)
        else:
            launch_apply(apply_jobs, reduce_jobs)

    def update_ddpg(self, A: ActorState, C: CriticState, batch: torch.Tensor):
        """`train_DDPG`'s update (Curious_GAN.py:271-281): extrinsic reward column, no G/D."""
        S, Ad = self.s_dim, self.a_dim
        b_s, b_s_ = batch[:, :S], batch[:, S:2 * S]
        b_a, b_r = batch[:, 3 * S:3 * S + Ad], batch[:, 3 * S + Ad]
        self.critic_step(C, A, b_s, b_a, b_r, b_s_)
        self.actor_step(A, C, b_s)
