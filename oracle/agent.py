"""Oracle (test infrastructure): the graph executions of the reference's learning and acting path.

Restates, in NumPy, what each `sess.run` of the reference computes:

  critic.learn      DDPG.py:141-151,174-178   (called at Curious_GAN.py:208)
  actor.learn       DDPG.py:97-113            (Curious_GAN.py:209)
  discriminator.learn / eval / determine      Curious_GAN.py:33-45,84-95   (:210,:212,:194)
  generator.learn / eval / predict            Curious_GAN.py:137-153       (:211,:213,:192)
  actor.choose_action                         DDPG.py:103-105              (:190)

Semantics restated from TensorFlow 1.x (parity unpinned, TF not installable here):
  * `RMSPropOptimizer(lr).minimize(loss)` in `Critic.__init__` has no `var_list`
    (DDPG.py:148): it differentiates the TD loss with respect to every trainable variable
    the loss reaches.  `q` is built on the tensor `actor.a`, so the actor's eval-net
    variables are reached; feeding `a` (DDPG.py:175) replaces the VALUE of that tensor
    but not the gradient graph.  The critic's train-op therefore also updates the actor
    eval net, with TanhGrad evaluated at the fed action, and with RMSProp slots of its
    own for those variables.  `critic_trains_actor=False` turns this off.
  * `tf.gradients(q, a)` is the gradient of sum_b q_b (no 1/B); `tf.gradients(a, params,
    grad_ys)` sums over the batch (DDPG.py:109,151).
  * target nets are hard-copied from the (already updated) eval nets when
    `counter % t_replace_iter == 0`, counter starting at 0 (DDPG.py:99-101,176-178).

`n_shards`/`global_batch`: the data-parallel contract -- mean-type losses are divided by
the GLOBAL batch so that summing shard gradients reproduces the single-process result.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import nets
from .optim import Adam, RMSProp


@dataclass
class Hyper:
    # Curious_GAN.py:325-339
    lr_a: float = 1e-4
    lr_c: float = 1e-4
    lr_d: float = 1e-4
    lr_g: float = 1e-4
    gamma: float = 0.9
    replace_iter_a: int = 1100
    replace_iter_c: int = 1000
    ita: float = 0.01
    smooth: float = 0.0
    critic_trains_actor: bool = True


class CuriousAgent:
    """Actor, Critic (eval + target), Generator, Discriminator and their optimizers."""

    def __init__(self, seed: int = 1234, dtype=np.float64, hyper: Optional[Hyper] = None,
                 s_dim: int = nets.STATE_DIM, a_dim: int = nets.ACTION_DIM):
        self.dtype = dtype
        self.hp = hyper or Hyper()
        self.actor_spec = nets.actor_spec(s_dim, a_dim)
        self.critic_spec = nets.critic_spec(s_dim, a_dim)
        self.gen_spec = nets.generator_spec(s_dim, a_dim)
        self.disc_spec = nets.discriminator_spec(s_dim, a_dim)
        rng = np.random.RandomState(seed)
        # creation order of the reference graph (Curious_GAN.py:355-360): actor eval, actor
        # target, critic eval, critic target, generator, discriminator -- each target net
        # gets its OWN draw (DDPG.py:70-73,133-136).
        self.actor_e = nets.init_params(self.actor_spec, rng, dtype)
        self.actor_t = nets.init_params(self.actor_spec, rng, dtype)
        self.critic_e = nets.init_params(self.critic_spec, rng, dtype)
        self.critic_t = nets.init_params(self.critic_spec, rng, dtype)
        self.gen = nets.init_params(self.gen_spec, rng, dtype)
        self.disc = nets.init_params(self.disc_spec, rng, dtype)
        na, nc = self.actor_spec.n_params, self.critic_spec.n_params
        self.opt_c = RMSProp(nc, self.hp.lr_c, dtype)          # C_train slots, critic vars
        self.opt_c_actor = RMSProp(na, self.hp.lr_c, dtype)    # C_train slots, actor vars
        self.opt_a = RMSProp(na, -self.hp.lr_a, dtype)         # A_train, ascent (DDPG.py:112)
        self.opt_d = Adam(self.disc_spec.n_params, self.hp.lr_d, dtype)
        self.opt_g = Adam(self.gen_spec.n_params, self.hp.lr_g, dtype)
        self.counter_a = 0
        self.counter_c = 0

    # ---- helpers -------------------------------------------------------
    def _c(self, x):
        return np.asarray(x, dtype=self.dtype)

    # ---- acting path ---------------------------------------------------
    def choose_action(self, s):
        a, _ = nets.forward(self.actor_spec, self.actor_e, {"s": self._c(s).reshape(-1, self.actor_spec.first[0][2])})
        return a

    def predict(self, s, a):
        g, _ = nets.forward(self.gen_spec, self.gen, {"s": self._c(s), "a": self._c(a)})
        return g

    def d_prob(self, s, a, x):
        logit, _ = nets.forward(self.disc_spec, self.disc, {"s": self._c(s), "x": self._c(x), "a": self._c(a)})
        return nets.sigmoid(logit)

    def determine(self, s, a, g):
        """1 - D(s,a,g)  (Curious_GAN.py:87-92)."""
        return self.dtype(1) - self.d_prob(s, a, g)

    def intrinsic_reward(self, r, s, a, g):
        """r + ITA*(1 - D(s,a,G))  (Curious_GAN.py:194)."""
        return self._c(r).reshape(-1, 1) + self.dtype(self.hp.ita) * self.determine(s, a, g)

    # ---- (1) critic.learn ----------------------------------------------
    def critic_grads(self, s, a, r, s_, global_batch: Optional[int] = None) -> Dict:
        s, a, r, s_ = self._c(s), self._c(a), self._c(r).reshape(-1, 1), self._c(s_)
        B = s.shape[0]
        gb = B if global_batch is None else global_batch
        a_t, _ = nets.forward(self.actor_spec, self.actor_t, {"s": s_})
        q_t, _ = nets.forward(self.critic_spec, self.critic_t, {"s": s_, "a": a_t})
        y = r + self.dtype(self.hp.gamma) * q_t                      # DDPG.py:142
        q, cache = nets.forward(self.critic_spec, self.critic_e, {"s": s, "a": a})
        diff = y - q
        loss_sum = float(np.sum(diff * diff))                        # DDPG.py:145 (mean = /B)
        dq = (self.dtype(-2.0) / self.dtype(gb)) * diff
        g_c, d_in = nets.backward(self.critic_spec, self.critic_e, cache, dq)
        out = {"loss_sum": loss_sum, "loss": loss_sum / gb, "q": q, "y": y,
               "g_critic": nets.flatten(self.critic_spec, g_c), "da": d_in["a"]}
        if self.hp.critic_trains_actor:
            # gradient continues into Actor/eval_net through the tensor actor.a; tanh's
            # output is the FED action, the hidden activations come from S = s.
            _, a_cache = nets.forward(self.actor_spec, self.actor_e, {"s": s})
            g_a, _ = nets.backward(self.actor_spec, self.actor_e, a_cache, d_in["a"], out_override=a)
            out["g_actor"] = nets.flatten(self.actor_spec, g_a)
        return out

    def critic_apply(self, g_critic, g_actor=None):
        flat = nets.flatten(self.critic_spec, self.critic_e)
        self.critic_e = nets.unflatten(self.critic_spec, self.opt_c.step(flat, g_critic))
        if g_actor is not None:
            fa = nets.flatten(self.actor_spec, self.actor_e)
            self.actor_e = nets.unflatten(self.actor_spec, self.opt_c_actor.step(fa, g_actor))
        if self.counter_c % self.hp.replace_iter_c == 0:
            self.critic_t = nets.unflatten(self.critic_spec, nets.flatten(self.critic_spec, self.critic_e))
        self.counter_c += 1

    def critic_learn(self, s, a, r, s_):
        out = self.critic_grads(s, a, r, s_)
        self.critic_apply(out["g_critic"], out.get("g_actor"))
        return out

    # ---- (2) actor.learn -----------------------------------------------
    def actor_grads(self, s) -> Dict:
        s = self._c(s)
        a, a_cache = nets.forward(self.actor_spec, self.actor_e, {"s": s})
        q, c_cache = nets.forward(self.critic_spec, self.critic_e, {"s": s, "a": a})
        _, d_in = nets.backward(self.critic_spec, self.critic_e, c_cache, np.ones_like(q))
        a_grads = d_in["a"]                                          # DDPG.py:151, no 1/B
        g_a, _ = nets.backward(self.actor_spec, self.actor_e, a_cache, a_grads)  # DDPG.py:109
        return {"a": a, "q": q, "a_grads": a_grads, "g_actor": nets.flatten(self.actor_spec, g_a)}

    def actor_apply(self, g_actor):
        fa = nets.flatten(self.actor_spec, self.actor_e)
        self.actor_e = nets.unflatten(self.actor_spec, self.opt_a.step(fa, g_actor))
        if self.counter_a % self.hp.replace_iter_a == 0:
            self.actor_t = nets.unflatten(self.actor_spec, nets.flatten(self.actor_spec, self.actor_e))
        self.counter_a += 1

    def actor_learn(self, s):
        out = self.actor_grads(s)
        self.actor_apply(out["g_actor"])
        return out

    # ---- (3) discriminator.learn / (5) eval ------------------------------
    def disc_loss_parts(self, g_data, s_, s, a):
        g_data, s_, s, a = self._c(g_data), self._c(s_), self._c(s), self._c(a)
        lf, cache_f = nets.forward(self.disc_spec, self.disc, {"s": s, "x": g_data, "a": a})
        lr, cache_r = nets.forward(self.disc_spec, self.disc, {"s": s, "x": s_, "a": a})
        ce_f = nets.sigmoid_ce(lf, 0.0)                               # Curious_GAN.py:33-35
        ce_r = nets.sigmoid_ce(lr, 1.0 - self.hp.smooth)              # Curious_GAN.py:37-39
        return lf, lr, cache_f, cache_r, ce_f, ce_r

    def disc_eval(self, g_data, s_, s, a, global_batch: Optional[int] = None) -> float:
        _, _, _, _, ce_f, ce_r = self.disc_loss_parts(g_data, s_, s, a)
        gb = ce_f.shape[0] if global_batch is None else global_batch
        return float(ce_r.sum() / gb + ce_f.sum() / gb)              # Curious_GAN.py:40

    def disc_grads(self, g_data, s_, s, a, global_batch: Optional[int] = None) -> Dict:
        lf, lr, cache_f, cache_r, ce_f, ce_r = self.disc_loss_parts(g_data, s_, s, a)
        B = lf.shape[0]
        gb = B if global_batch is None else global_batch
        d_f = (nets.sigmoid(lf) - self.dtype(0.0)) / self.dtype(gb)
        d_r = (nets.sigmoid(lr) - self.dtype(1.0 - self.hp.smooth)) / self.dtype(gb)
        g_f, _ = nets.backward(self.disc_spec, self.disc, cache_f, d_f)
        g_r, _ = nets.backward(self.disc_spec, self.disc, cache_r, d_r)
        g = nets.flatten(self.disc_spec, g_f) + nets.flatten(self.disc_spec, g_r)
        return {"loss": float(ce_r.sum() / gb + ce_f.sum() / gb), "g_disc": g,
                "logit_fake": lf, "logit_real": lr}

    def disc_apply(self, g_disc):
        fd = nets.flatten(self.disc_spec, self.disc)
        self.disc = nets.unflatten(self.disc_spec, self.opt_d.step(fd, g_disc))

    def disc_learn(self, g_data, s_, s, a):
        out = self.disc_grads(g_data, s_, s, a)
        self.disc_apply(out["g_disc"])
        return out

    # ---- (4) generator.learn / (6) eval ----------------------------------
    def gen_eval(self, s, a, global_batch: Optional[int] = None) -> float:
        s, a = self._c(s), self._c(a)
        g, _ = nets.forward(self.gen_spec, self.gen, {"s": s, "a": a})
        lf, _ = nets.forward(self.disc_spec, self.disc, {"s": s, "x": g, "a": a})
        gb = lf.shape[0] if global_batch is None else global_batch
        return float(nets.sigmoid_ce(lf, 1.0).sum() / gb)            # Curious_GAN.py:139-141

    def gen_grads(self, s, a, global_batch: Optional[int] = None) -> Dict:
        s, a = self._c(s), self._c(a)
        g, g_cache = nets.forward(self.gen_spec, self.gen, {"s": s, "a": a})
        lf, d_cache = nets.forward(self.disc_spec, self.disc, {"s": s, "x": g, "a": a})
        B = lf.shape[0]
        gb = B if global_batch is None else global_batch
        loss = float(nets.sigmoid_ce(lf, 1.0).sum() / gb)
        d_logit = (nets.sigmoid(lf) - self.dtype(1.0)) / self.dtype(gb)
        _, d_in = nets.backward(self.disc_spec, self.disc, d_cache, d_logit)
        g_g, _ = nets.backward(self.gen_spec, self.gen, g_cache, d_in["x"])
        return {"loss": loss, "g_gen": nets.flatten(self.gen_spec, g_g), "g": g, "dg": d_in["x"]}

    def gen_apply(self, g_gen):
        fg = nets.flatten(self.gen_spec, self.gen)
        self.gen = nets.unflatten(self.gen_spec, self.opt_g.step(fg, g_gen))

    def gen_learn(self, s, a):
        out = self.gen_grads(s, a)
        self.gen_apply(out["g_gen"])
        return out

    # ---- one full update (Curious_GAN.py:199-213) -----------------------
    def update(self, batch: np.ndarray, s_dim: int = nets.STATE_DIM, a_dim: int = nets.ACTION_DIM) -> Dict:
        """`batch` is [B, 3*S+A+2] in the 6-field replay layout (Curious_GAN.py:162,200-205)."""
        b_s = batch[:, :s_dim]
        b_s_ = batch[:, s_dim:2 * s_dim]
        b_g = batch[:, 2 * s_dim:3 * s_dim]
        b_a = batch[:, -2 - a_dim:-2]
        b_cr = batch[:, -1:]
        c = self.critic_learn(b_s, b_a, b_cr, b_s_)
        a = self.actor_learn(b_s)
        d = self.disc_learn(b_g, b_s_, b_s, b_a)
        g = self.gen_learn(b_s, b_a)
        d_loss = self.disc_eval(b_g, b_s_, b_s, b_a)
        g_loss = self.gen_eval(b_s, b_a)
        return {"critic": c, "actor": a, "disc": d, "gen": g,
                "critic_loss": c["loss"], "D_loss": d_loss, "G_loss": g_loss}

    def update_ddpg(self, batch: np.ndarray, s_dim: int = nets.STATE_DIM, a_dim: int = nets.ACTION_DIM) -> Dict:
        """`train_DDPG` (Curious_GAN.py:271-281): extrinsic reward column -2, no G/D."""
        b_s = batch[:, :s_dim]
        b_s_ = batch[:, s_dim:2 * s_dim]
        b_a = batch[:, -2 - a_dim:-2]
        b_r = batch[:, -2:-1]
        c = self.critic_learn(b_s, b_a, b_r, b_s_)
        a = self.actor_learn(b_s)
        return {"critic": c, "actor": a, "critic_loss": c["loss"]}

    # ---- flat views ------------------------------------------------------
    def flat(self, which: str) -> np.ndarray:
        spec = {"actor_e": self.actor_spec, "actor_t": self.actor_spec, "critic_e": self.critic_spec,
                "critic_t": self.critic_spec, "gen": self.gen_spec, "disc": self.disc_spec}[which]
        return nets.flatten(spec, getattr(self, which))
