"""Helpers shared by the GPU parity tests: build an oracle agent and device networks holding the
same weights, and compare with the tolerance the build is held to.

Tolerance: |got - ref| <= RTOL*|ref| + RTOL*max|ref| per tensor, RTOL = 1e-5 -- "within 1e-5
relative (fp32)" with the per-tensor scale as the floor for entries that cancel towards zero
(the float64 oracle is the reference value; the device computes in fp32).
"""
import numpy as np
import torch

import rl_gan_amd as R
from rl_gan_amd import _lib
from rl_gan_amd.core import DeviceNet
from rl_gan_amd.engine import ActorState, AdamState, CriticState, StepEngine
from oracle import nets as onets
from oracle.agent import CuriousAgent, Hyper
from oracle_util import load_weights, make_batch, make_oracle, make_smooth_batch  # noqa: F401

RTOL = 1e-5


def close(got, ref, rtol=RTOL, what="", atol=0.0):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = np.max(np.abs(ref)) if ref.size else 0.0
    err = np.abs(got - ref)
    tol = rtol * np.abs(ref) + rtol * scale + atol
    bad = err > tol
    if bad.any():
        i = np.argmax(err - tol)
        raise AssertionError(f"{what}: {bad.sum()}/{bad.size} outside tolerance; worst idx {i}: got {got.flat[i]!r} "
                             f"ref {ref.flat[i]!r} err {err.flat[i]:.3e} tol {tol.flat[i]:.3e} scale {scale:.3e}")


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).cuda()


class Pair:
    """An oracle agent (float64) and the device states with identical weights."""

    def __init__(self, seed=1234, hyper=None, spread=True, weights=None):
        self.hp = hyper or Hyper()
        self.o = make_oracle(seed, self.hp, spread)
        if weights is not None:
            load_weights(self.o, weights)
        o = self.o
        mk = lambda kind, spec, params, train: DeviceNet(kind, 7, 2, trainable=train, init=onets.flatten(spec, params))
        self.A = ActorState(e=mk(_lib.NET_ACTOR, o.actor_spec, o.actor_e, True), t=mk(_lib.NET_ACTOR, o.actor_spec, o.actor_t, False),
                            lr=self.hp.lr_a, period=self.hp.replace_iter_a)
        self.C = CriticState(e=mk(_lib.NET_CRITIC, o.critic_spec, o.critic_e, True), t=mk(_lib.NET_CRITIC, o.critic_spec, o.critic_t, False),
                             lr=self.hp.lr_c, gamma=self.hp.gamma, period=self.hp.replace_iter_c,
                             trains_actor=self.hp.critic_trains_actor)
        self.G = AdamState(net=mk(_lib.NET_GENERATOR, o.gen_spec, o.gen, True), lr=self.hp.lr_g)
        self.D = AdamState(net=mk(_lib.NET_DISCRIMINATOR, o.disc_spec, o.disc, True), lr=self.hp.lr_d)
        self.eng = StepEngine(7, 2)

    def check_flats(self, flats, what="", adam_steps=None):
        """Six flat weight vectors (dict by name) against the oracle's.  RMSProp-trained networks: the plain
        tolerance.  Adam-trained ones (generator, discriminator): plus, per entry, the bound the oracle's optimizer
        keeps on what gradient errors within RTOL may do to that entry (oracle/optim.py, Adam.allow) -- about a
        step where the gradient is rounding noise, next to nothing where it is not."""
        o = self.o
        for k in ("actor_e", "actor_t", "critic_e", "critic_t"):
            close(flats[k], o.flat(k), what=f"{what} {k}")
        close(flats["gen"], o.flat("gen"), what=what + " gen", atol=o.opt_g.allow)
        close(flats["disc"], o.flat("disc"), what=what + " disc", atol=o.opt_d.allow)

    def check_weights(self, what="", adam_steps=None):
        self.check_flats({"actor_e": self.A.e.get_flat(), "actor_t": self.A.t.get_flat(), "critic_e": self.C.e.get_flat(),
                          "critic_t": self.C.t.get_flat(), "gen": self.G.net.get_flat(), "disc": self.D.net.get_flat()}, what)

    def check_slots(self, what="", rtol=RTOL):
        """Optimizer slots against the oracle's: RMSProp's mean square (both of the critic's train-op and the
        actor's own), Adam's two moments.  They are linear / quadratic in the gradients, so after a first update from
        identical weights the plain tolerance holds.  Later updates start from weights that already differ where Adam
        took a noise-decided step (see check_flats), which moves every later gradient a little: callers pass a wider
        rtol there and say what they measured."""
        o = self.o
        close(self.C.ms.cpu().numpy(), o.opt_c.ms, what=what + " critic ms", rtol=rtol)
        if self.C.ms_actor is not None:
            close(self.C.ms_actor.cpu().numpy(), o.opt_c_actor.ms, what=what + " critic train-op actor ms", rtol=rtol)
        close(self.A.ms.cpu().numpy(), o.opt_a.ms, what=what + " actor ms", rtol=rtol)
        for st, opt, n in ((self.D, o.opt_d, "disc"), (self.G, o.opt_g, "gen")):
            close(st.m.cpu().numpy(), opt.m, what=f"{what} {n} m", rtol=rtol)
            close(st.v.cpu().numpy(), opt.v, what=f"{what} {n} v", rtol=2 * rtol)     # v is quadratic in g: twice its relative error
            assert int(st.step.item()) == opt.t

    def check_deltas(self, before, what=""):
        """The applied updates themselves, w_after - w_before, against the oracle's: the weights are O(0.1) and a step
        is O(lr), so comparing weights alone would let the optimizer arithmetic be off by percents.  Tolerance: RTOL of
        the update (with the tensor's largest update as floor) plus one float32 ulp of the weight, which is what
        storing w in float32 costs; Adam-trained networks get the oracle's gradient-noise bound on top."""
        o = self.o
        now = {"actor_e": self.A.e.get_flat(), "critic_e": self.C.e.get_flat(), "gen": self.G.net.get_flat(), "disc": self.D.net.get_flat()}
        for k, w in now.items():
            w0 = np.asarray(before[k], dtype=np.float64)
            ref = o.flat(k) - w0
            ulp = np.spacing(np.abs(w).astype(np.float32)).astype(np.float64)
            extra = {"gen": o.opt_g.allow, "disc": o.opt_d.allow}.get(k, 0.0)
            close(w.astype(np.float64) - w0, ref, what=f"{what} delta {k}", atol=ulp + extra)

    def snapshot(self):
        return {"actor_e": self.A.e.get_flat(), "critic_e": self.C.e.get_flat(), "gen": self.G.net.get_flat(), "disc": self.D.net.get_flat()}
