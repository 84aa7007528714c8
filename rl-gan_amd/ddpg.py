"""Actor, Critic and the 4-field replay Memory with the reference's API (GAN-RL/DDPG.py:57-196).

Constructor argument ORDER follows the reference's call sites (Curious_GAN.py:355-356,365-366),
which is what a caller of the reference actually writes.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib
from .core import DeviceNet, device, mlp_forward
from .engine import ActorState, CriticState
from .replay import DeviceMemory, device_view
from .session import Placeholder, Session, TensorHandle, init_rng


def _dev(x, width: Optional[int] = None) -> torch.Tensor:
    """float64/float32 host array (or device tensor) -> [B, width] fp32 device tensor: the cast TF
    applies when a float64 batch is fed to a float32 placeholder."""
    if isinstance(x, torch.Tensor):
        t = x.to(device=device(), dtype=torch.float32)
    elif (t := device_view(x)) is not None:                 # a slice of the minibatch `Memory.sample` just returned
        return t
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(device())
    if t.dim() == 1:
        t = t.reshape(-1, width) if width else t.reshape(-1, 1)
    return t.contiguous() if t.stride(-1) != 1 else t


class Actor(object):
    """DDPG.py:57-113."""

    def __init__(self, sess: Session, action_dim: int, action_bound, learning_rate: float, t_replace_iter: int,
                 S: Placeholder = None, S_: Placeholder = None, scope: str = 'Actor'):
        self.sess = sess
        self.a_dim = action_dim
        self.action_bound = action_bound
        self.lr = learning_rate
        self.t_replace_iter = t_replace_iter
        self.scope = scope
        self.S, self.S_ = S, S_
        self.s_dim = S.width
        rng = init_rng()
        e = DeviceNet(_lib.NET_ACTOR, self.s_dim, action_dim, trainable=True, rng=rng)    # eval_net
        t = DeviceNet(_lib.NET_ACTOR, self.s_dim, action_dim, trainable=False, rng=rng)   # target_net: its own draw
        self.state = ActorState(e=e, t=t, lr=learning_rate, period=t_replace_iter)
        self.a = TensorHandle(self, 'a')        # eval_net(S)
        self.a_ = TensorHandle(self, 'a_')      # target_net(S_)
        self.e_params = [f"{scope}/eval_net/{n}" for n, _ in e.table]
        self.t_params = [f"{scope}/target_net/{n}" for n, _ in t.table]
        self._critic = None

    @property
    def t_replace_counter(self) -> int:
        return int(self.state.counter.item())

    def learn(self, s):   # batch update, DDPG.py:97-101
        if self._critic is None:
            raise RuntimeError("call add_grad_to_graph(critic.a_grads) first (DDPG.py:107)")
        eng = self.sess.engine(self.s_dim, self.a_dim)
        eng.actor_step(self.state, self._critic.state, _dev(s, self.s_dim))

    def choose_action(self, s):   # DDPG.py:103-105
        s = np.asarray(s)[np.newaxis, :]
        return mlp_forward(self.state.e, _dev(s)).cpu().numpy()[0]

    def choose_action_batch(self, s) -> torch.Tensor:
        return mlp_forward(self.state.e, _dev(s, self.s_dim))

    def add_grad_to_graph(self, a_grads: TensorHandle):   # DDPG.py:107-113
        assert a_grads.role == 'a_grads'
        self._critic = a_grads.owner


class Critic(object):
    """DDPG.py:116-178.  `critic_trains_actor=True` keeps the behaviour of `minimize(loss)` without a
    var_list (DDPG.py:148): the critic's train-op also moves the actor's eval net."""

    def __init__(self, sess: Session, state_dim: int, action_dim: int, learning_rate: float, gamma: float,
                 t_replace_iter: int, a: TensorHandle, a_: TensorHandle, S: Placeholder = None, S_: Placeholder = None,
                 R: Placeholder = None, scope: str = 'Critic', critic_trains_actor: bool = True):
        self.sess = sess
        self.s_dim, self.a_dim = state_dim, action_dim
        self.lr, self.gamma, self.t_replace_iter = learning_rate, gamma, t_replace_iter
        self.a, self.a_, self.S, self.S_, self.R = a, a_, S, S_, R
        self.scope = scope
        rng = init_rng()
        e = DeviceNet(_lib.NET_CRITIC, state_dim, action_dim, trainable=True, rng=rng)
        t = DeviceNet(_lib.NET_CRITIC, state_dim, action_dim, trainable=False, rng=rng)
        self.state = CriticState(e=e, t=t, lr=learning_rate, gamma=gamma, period=t_replace_iter,
                                 trains_actor=critic_trains_actor)
        self.e_params = [f"{scope}/eval_net/{n}" for n, _ in e.table]
        self.t_params = [f"{scope}/target_net/{n}" for n, _ in t.table]
        self.q = TensorHandle(self, 'q')
        self.q_ = TensorHandle(self, 'q_')
        self.loss = TensorHandle(self, 'loss')
        self.a_grads = TensorHandle(self, 'a_grads')   # DDPG.py:151

    @property
    def t_replace_counter(self) -> int:
        return int(self.state.counter.item())

    def learn(self, s, a, r, s_):   # DDPG.py:174-178
        eng = self.sess.engine(self.s_dim, self.a_dim)
        actor = self.a.owner
        eng.critic_step(self.state, actor.state, _dev(s, self.s_dim), _dev(a, self.a_dim), _dev(r, 1), _dev(s_, self.s_dim))

    def last_loss(self) -> float:
        eng = self.sess.engine(self.s_dim, self.a_dim)
        return float(eng.losses[eng.LOSS_CRITIC].item())


class Memory(DeviceMemory):
    """DDPG.py:181-196: rows are [s, a, r, s_]."""

    def store_transition(self, s, a, r, s_):
        self.store_row(np.hstack((s, a, [r], s_)))
