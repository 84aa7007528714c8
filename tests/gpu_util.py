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
from oracle_util import load_weights, make_batch, make_oracle  # noqa: F401

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
