"""Generator, Discriminator and the 6-field replay Memory with the reference's API
(GAN-RL/Curious_GAN.py:11-169)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .core import DeviceNet, mlp_forward
from .ddpg import _dev
from .engine import AdamState
from .replay import DeviceMemory
from .session import Placeholder, Session, TensorHandle, init_rng


class Discriminator(object):
    """Curious_GAN.py:11-95.  `smooth` is a module global in the reference (Curious_GAN.py:39,319)."""

    def __init__(self, sess: Session, state_dim: int, action_dim: int, Learning_rate: float, a: TensorHandle,
                 G: TensorHandle, S_: Placeholder = None, S: Placeholder = None, smooth: float = 0.0):
        self.sess = sess
        self.s_dim, self.a_dim, self.G_dim = state_dim, action_dim, state_dim
        self.lr = Learning_rate
        self.a, self.G, self.S, self.S_ = a, G, S, S_
        self.smooth = smooth
        net = DeviceNet(_lib.NET_DISCRIMINATOR, state_dim, action_dim, trainable=True, rng=init_rng())
        self.state = AdamState(net=net, lr=Learning_rate)
        self.params = [f"Discriminator/eval_net/{n}" for n, _ in net.table]
        self.D_fake = TensorHandle(self, 'D_fake')
        self.D_logit_fake = TensorHandle(self, 'D_logit_fake')
        self.D_real = TensorHandle(self, 'D_real')
        self.D_logit_real = TensorHandle(self, 'D_logit_real')

    def _engine(self):
        return self.sess.engine(self.s_dim, self.a_dim)

    def learn(self, G_data, s_, s, a):   # Curious_GAN.py:84-85
        S, A = self.s_dim, self.a_dim
        self._engine().disc_step(self.state, _dev(G_data, S), _dev(s_, S), _dev(s, S), _dev(a, A), self.smooth)

    def determine(self, s, a, g):   # Curious_GAN.py:87-92 -> array of shape (1, 1)
        s, a, g = (np.asarray(v)[np.newaxis, :] for v in (s, a, g))
        return self.determine_batch(s, a, g).cpu().numpy()

    def determine_batch(self, s, a, g) -> torch.Tensor:
        """1 - D(s, a, g) for a batch (naming after Conv_model.py:296)."""
        S, A = self.s_dim, self.a_dim
        return mlp_forward(self.state.net, _dev(s, S), a=_dev(a, A), x=_dev(g, S), out_mode=2)

    def eval(self, G_data, s_, s, a):   # Curious_GAN.py:94-95
        S, A = self.s_dim, self.a_dim
        gen = self.G.owner
        out = self._engine().gan_eval(gen.state, self.state, _dev(G_data, S), _dev(s_, S), _dev(s, S), _dev(a, A), self.smooth)
        return float(out[0].item())


class Generator(object):
    """Curious_GAN.py:97-153."""

    def __init__(self, sess: Session, state_dim: int, action_dim: int, Learning_rate: float, a: TensorHandle,
                 S: Placeholder = None):
        self.sess = sess
        self.s_dim, self.a_dim, self.G_dim = state_dim, action_dim, state_dim
        self.lr = Learning_rate
        self.a, self.S = a, S
        net = DeviceNet(_lib.NET_GENERATOR, state_dim, action_dim, trainable=True, rng=init_rng())
        self.state = AdamState(net=net, lr=Learning_rate)
        self.vars = [f"Generator/eval_net/{n}" for n, _ in net.table]
        self.G = TensorHandle(self, 'G')
        self._disc = None

    def _engine(self):
        return self.sess.engine(self.s_dim, self.a_dim)

    def model_loss(self, D_logit_fake: TensorHandle):   # Curious_GAN.py:137-142
        assert D_logit_fake.role == 'D_logit_fake'
        self._disc = D_logit_fake.owner

    def learn(self, s, a):   # Curious_GAN.py:144-145
        if self._disc is None:
            raise RuntimeError("call model_loss(discriminator.D_logit_fake) first (Curious_GAN.py:362)")
        self._engine().gen_step(self.state, self._disc.state, _dev(s, self.s_dim), _dev(a, self.a_dim))

    def eval(self, s, a):   # Curious_GAN.py:147-148
        S, A = self.s_dim, self.a_dim
        s, a = _dev(s, S), _dev(a, A)
        # the G loss does not read the stored sample or s_: any finite rows do for the D half
        out = self._engine().gan_eval(self.state, self._disc.state, s, s, s, a, self._disc.smooth)
        return float(out[1].item())

    def predict(self, s, a):   # Curious_GAN.py:150-153
        s, a = np.asarray(s)[np.newaxis, :], np.asarray(a)[np.newaxis, :]
        return self.predict_batch(s, a).cpu().numpy()[0]

    def predict_batch(self, s, a) -> torch.Tensor:
        """G(s, a) for a batch (naming after Conv_model.py:562)."""
        return mlp_forward(self.state.net, _dev(s, self.s_dim), a=_dev(a, self.a_dim))


class Memory(DeviceMemory):
    """Curious_GAN.py:155-169: rows are [s, s_, G, a, r, curious_r]."""

    def store_transition(self, s, s_, G, a, r, curious_r):
        self.store_row(np.hstack((s, s_, G, a, [r], curious_r)))
