"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): one learning iteration of the pixel variant, composed.

Restates, in float64 NumPy, the calls the pixel training loop makes on one minibatch (Curious_GAN_mario.py:86-98)
in the reference's order, on the step functions of oracle/pixel_gan.py, oracle/inverse_model.py and
oracle/pixel_ddpg.py:

    generator.learn(b_s, b_a, lr, b_s_)                    Adam on the generator's variables        Conv_model.py:531-541
    b_g = generator.predict_batch(b_s, b_a)                with the generator as just updated       Conv_model.py:562-567
    discriminator.learn(b_g, b_s_[..., [-1]], b_s, b_a)    Adam on D + state encoder + frame encoder Conv_model.py:207-223,270-281
    A_pred.learn(b_s, b_s_, b_a)                           Adam on A_pred + state encoder           Conv_model.py:811-820,864-867
    critic.learn(b_s, ., b_a, b_curious_r, b_s_, .)        RMSProp(lr), target copy on its period   Conv_model.py:610-616,653-663
    actor.learn(b_s, .)                                    RMSProp(-lr) on sum_b da/dtheta dq/da    Conv_model.py:746-753,778-784

What the composition adds to the single steps: the state encoder is trained by two optimizers with slots of their
own (every tf.train.AdamOptimizer instance creates its own m and v for a variable it minimises), the frame encoder by
one while the generator's loss only passes through it, the critic and the actor read the encoder as the two steps
before left it, the actor's step sees the critic's update of the same iteration, and each target network follows its
own counter.  USE_BATCH_NORM and USE_LSTM are off as in the reference's script (Curious_GAN_mario.py:384,
Conv_model.py:667).

Parity unpinned against TensorFlow (not installed; the reference holds no fixture for these graphs)."""
import numpy as np

from . import conv as oc
from . import deconv as odc
from . import inverse_model as om
from . import pixel_ddpg as od
from . import pixel_gan as op
from .optim import Adam, RMSProp


class _Slots:
    """One optimizer instance: a slot object per variable, created at first use like TensorFlow's."""

    def __init__(self, make):
        self.make, self.slots = make, {}

    def step(self, name, var, grad):
        if name not in self.slots:
            self.slots[name] = self.make(var.size)
        o = self.slots[name]
        return o.step(var.ravel(), np.asarray(grad, dtype=np.float64).ravel()).reshape(var.shape)


def _enc_step(opt, prefix, params, dense, grads):
    """Applies [(dw, db) per convolution] + [(dw, db) of the 256-unit layer] to an encoder; returns the new pair."""
    new = []
    for i, ((w, b), (dw, db)) in enumerate(zip(params, grads[:-1])):
        new.append((opt.step(f"{prefix}.conv{i}.w", w, dw), opt.step(f"{prefix}.conv{i}.b", b, db)))
    dwd, dbd = grads[-1]
    return new, (opt.step(prefix + ".dense.w", dense[0], dwd), opt.step(prefix + ".dense.b", dense[1], dbd))


def _dict_step(opt, prefix, P, G, keys=None):
    return {k: (opt.step(f"{prefix}.{k}", P[k], np.asarray(G[k]).reshape(P[k].shape)) if (keys is None or k in keys) else P[k])
            for k in P}


class PixelIteration:
    """Holds every variable of the pixel variant's learner as float64 arrays shaped like the reference's and advances
    them one minibatch at a time.  All arguments are taken over as they are (no copies are shared with the caller
    afterwards: every step builds new arrays).

        st_params, st_dense    state encoder: [(kernel [3,3,4,32], bias), (kernel [3,3,32,32], bias)], (kernel, bias)
        enc_params, enc_dense  single-frame encoder, first kernel [3,3,1,32]
        gen                    dict w1_s, w1_a, b1, decoder = [(kernel, bias) x 4]
        disc                   dict w1_G, w1_s, w1_a, b1, w3, b3, w2, b2
        a_pred                 dict of oracle/inverse_model.py
        actor, actor_t, critic, critic_t   dicts of oracle/pixel_ddpg.py"""

    def __init__(self, st_params, st_dense, enc_params, enc_dense, gen, disc, a_pred, actor, actor_t, critic, critic_t,
                 lr_g=1e-4, lr_d=1e-4, lr_a_pred=1e-4, lr_c=1e-4, lr_a=1e-4, gamma=0.99, period_c=1000, period_a=1100):
        self.st_params, self.st_dense, self.enc_params, self.enc_dense = list(st_params), tuple(st_dense), list(enc_params), tuple(enc_dense)
        self.gen, self.disc, self.a_pred = dict(gen), dict(disc), dict(a_pred)
        self.actor, self.actor_t, self.critic, self.critic_t = dict(actor), dict(actor_t), dict(critic), dict(critic_t)
        self.gamma, self.period_c, self.period_a = gamma, period_c, period_a
        self.counter_c = self.counter_a = 0
        self.opt_g = _Slots(lambda n: Adam(n, lr_g))
        self.opt_d = _Slots(lambda n: Adam(n, lr_d))
        self.opt_a_pred = _Slots(lambda n: Adam(n, lr_a_pred))
        self.opt_c = _Slots(lambda n: RMSProp(n, lr_c))
        self.opt_a = _Slots(lambda n: RMSProp(n, -lr_a))                 # ascent: RMSPropOptimizer(-lr), Conv_model.py:783

    def features(self, s):
        return op.encoder(s, self.st_params, self.st_dense, oc.universe_head(len(self.st_params)))[0]

    def predict_batch(self, phis, a):
        G = self.gen
        code = np.maximum(phis @ G["w1_s"] + a @ G["w1_a"] + G["b1"], 0)
        return odc.decoder_forward(code, G["decoder"])[0]

    def iteration(self, s, a, r, s_next):
        """s, s_next [n, 42, 42, 4] in [0, 1]; a [n, a_dim]; r [n, 1] (the loop feeds curious_r).  Returns the losses
        each step reports, measured before that step's update like the device's: G_loss, D_loss_real, D_loss_fake,
        A_pred_loss, critic_loss; and the image the discriminator saw."""
        out = {}
        last = s_next[:, :, :, -1:]
        # generator.learn: phis through the state encoder as it stands; only the generator's variables move
        phis = self.features(s)
        G = self.gen
        loss, _, g = op.generator_loss_and_grads(phis, a, last, G["w1_s"], G["w1_a"], G["b1"], G["decoder"], self.enc_params,
                                                 self.enc_dense)
        out["G_loss"] = loss
        new = _dict_step(self.opt_g, "G", {k: G[k] for k in ("w1_s", "w1_a", "b1")}, g)
        new["decoder"] = [(self.opt_g.step(f"G.dec{i}.w", w, dw), self.opt_g.step(f"G.dec{i}.b", b, db))
                          for i, ((w, b), (dw, db)) in enumerate(zip(G["decoder"], g["decoder"]))]
        self.gen = new
        # b_g from the updated generator, fed to the discriminator as data
        image = self.predict_batch(phis, a)
        out["image"] = image
        lr_, lf_, g = op.discriminator_loss_and_grads(s, a, last, image, self.disc, self.st_params, self.st_dense, self.enc_params,
                                                      self.enc_dense)
        out["D_loss_real"], out["D_loss_fake"] = lr_, lf_
        self.disc = _dict_step(self.opt_d, "D", self.disc, g)
        self.st_params, self.st_dense = _enc_step(self.opt_d, "st", self.st_params, self.st_dense, g["state_encoder"])
        self.enc_params, self.enc_dense = _enc_step(self.opt_d, "ob", self.enc_params, self.enc_dense, g["obs_encoder"])
        # inverse model: its own Adam, and its own slots for the state encoder
        loss, g = om.loss_and_grads(s, s_next, a, self.a_pred, self.st_params, self.st_dense)
        out["A_pred_loss"] = loss
        self.a_pred = _dict_step(self.opt_a_pred, "A_pred", self.a_pred, g)
        self.st_params, self.st_dense = _enc_step(self.opt_a_pred, "st", self.st_params, self.st_dense, g["state_encoder"])
        # critic, then actor, on the encoder as it is now; neither trains it
        phis, phis_next = self.features(s), self.features(s_next)
        loss, g = od.critic_loss_and_grads(phis, a, r, phis_next, self.critic, self.critic_t, self.actor_t, self.gamma)
        out["critic_loss"] = loss
        self.critic = _dict_step(self.opt_c, "C", self.critic, g)
        if self.counter_c % self.period_c == 0:
            self.critic_t = dict(self.critic)
        self.counter_c += 1
        _, _, g = od.actor_policy_grads(phis, self.actor, self.critic)
        self.actor = _dict_step(self.opt_a, "A", self.actor, g)
        if self.counter_a % self.period_a == 0:
            self.actor_t = dict(self.actor)
        self.counter_a += 1
        return out
