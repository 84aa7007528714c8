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

    def __init__(self, seed=1234, hyper=None, spread=True):
        self.hp = hyper or Hyper()
        self.o = CuriousAgent(seed=seed, dtype=np.float64, hyper=self.hp)
        if spread:
            # make clipped / negative-side activations common so every mask branch is exercised
            rng = np.random.RandomState(seed + 1)
            for p in (self.o.actor_e, self.o.actor_t, self.o.critic_e, self.o.critic_t, self.o.gen, self.o.disc):
                for k in p:
                    if k.startswith("l1/") and p[k].shape[0] > 1:
                        p[k] = p[k] * 6.0
                    if "bias" in k or k.endswith("b1") or k.endswith("b3"):
                        p[k] = p[k] + rng.uniform(-0.3, 0.3, p[k].shape)
            self._round_all()
        else:
            self._round_all()
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

    def _round_all(self):
        # the device holds fp32 weights: start the float64 oracle from the same representable values
        for p in (self.o.actor_e, self.o.actor_t, self.o.critic_e, self.o.critic_t, self.o.gen, self.o.disc):
            for k in p:
                p[k] = p[k].astype(np.float32).astype(np.float64)

    def check_weights(self, what="", adam_steps=1):
        """Adam divides by sqrt(v): an entry whose gradient is rounding noise still moves by about
        lr per step, in a direction the noise decides.  Adam-trained weights therefore get an extra
        absolute allowance of a quarter of lr per step taken; gradients are compared strictly."""
        o = self.o
        ad = 0.25 * adam_steps
        close(self.A.e.get_flat(), o.flat("actor_e"), what=what + " actor_e")
        close(self.A.t.get_flat(), o.flat("actor_t"), what=what + " actor_t")
        close(self.C.e.get_flat(), o.flat("critic_e"), what=what + " critic_e")
        close(self.C.t.get_flat(), o.flat("critic_t"), what=what + " critic_t")
        close(self.G.net.get_flat(), o.flat("gen"), what=what + " gen", atol=ad * self.hp.lr_g)
        close(self.D.net.get_flat(), o.flat("disc"), what=what + " disc", atol=ad * self.hp.lr_d)


def make_batch(B, seed=0, f32=True):
    """[B,25] rows in the 6-field layout with plausible ranges (states ~[-1.5,1.5], actions [-1,1])."""
    rng = np.random.RandomState(seed)
    b = np.empty((B, 25))
    b[:, 0:21] = rng.uniform(-1.5, 1.5, (B, 21))
    b[:, 0] = rng.randint(0, 2, B)
    b[:, 21:23] = rng.uniform(-1, 1, (B, 2))
    b[:, 23] = np.where(rng.rand(B) < 0.1, 10.0, 0.0)
    b[:, 24] = b[:, 23] + 0.01 * rng.rand(B)
    if f32:
        b = b.astype(np.float32).astype(np.float64)
    return b
