"""Actor and critic of the reference's pixel variant (`Actor`, `Critic`, Conv_model.py:569-784) as
Curious_GAN_mario.py:319-325 wires them, with USE_BATCH_NORM = False (Curious_GAN_mario.py:384) and USE_LSTM = False
(Conv_model.py:667,701-706), on the 256 features `phis` of the state encoder:

    a = tanh(dense(relu6(dense10(relu6(dense200(relu6(dense200(phis))))))))                    Conv_model.py:711-744
    q = dense1(relu6(dense10(relu6(dense200(relu6(phis w1_s + a w1_a + b1))))))                Conv_model.py:621-651
    critic.learn: loss = mean((R + gamma q'(phis', a'(phis')) - q(phis, a))^2), RMSProp(lr) on the critic's eval
                  variables (var_list = e_params, Conv_model.py:610-616,653-663)
    actor.learn:  sum_b da/dtheta * dq/da through the eval critic, RMSProp(-lr) on the actor's eval dense variables
                  (Conv_model.py:619,746-754,778-784)
    both: hard copy eval -> target when counter % t_replace_iter == 0, after the step, counter from 0

Neither step trains the state encoder (it is in neither variable list).  The 10-unit layers are kept 12 wide and the
action / q layers a multiple of 4 wide (16-byte rows); the extra units have zero weights and receive zero gradients."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

import ctypes as C

from . import _lib
from ._lib import check, lib
from .conv import ConvEncoder, DenseLayer
from .core import (DeviceNet, Workspace, _stream, device, launch_apply, launch_wgrads, mat, reduce_job, rmsprop_job, row_ld, rows16,
                   wgrad_job)

N_L1, N_L2, N_L3, L3_PAD = 200, 200, 10, 12                     # Conv_model.py:627,636,643 / :716,723,730


def _pad(kernel, bias, k_pad: int, n_pad: int):
    k, b = np.zeros((k_pad, n_pad)), np.zeros(n_pad)
    kernel, bias = np.asarray(kernel), np.asarray(bias).reshape(-1)
    k[:kernel.shape[0], :kernel.shape[1]], b[:bias.size] = kernel, bias
    return k, b


def _xavier(rng, i: int, o: int) -> np.ndarray:
    lim = np.sqrt(6.0 / (i + o))
    return rng.uniform(-lim, lim, (i, o))


class _RMSProp:
    """tf.train.RMSPropOptimizer slot of one layer (ms starts at one), with the target layer it copies to."""

    def __init__(self, layer: DenseLayer, target: DenseLayer):
        self.layer, self.target = layer, target
        self.ms = torch.ones_like(layer.canon)

    def job(self, lr: float, counter: torch.Tensor, period: int) -> _lib.ApplyJob:
        L = self.layer
        return _lib.ApplyJob(0, L.n_params, L.canon.data_ptr(), L.grad.data_ptr(), self.ms.data_ptr(), None, lr, 0.9, 0.0, 1e-10,
                             None, None, None, None, None, self.target.canon.data_ptr(), None, 1.0, counter.data_ptr(), period, None)


class _Net:
    """Four dense layers used from their canonical parameters."""

    def __init__(self, layers: List[DenseLayer], acts):
        self.layers, self.acts = layers, acts

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        """[x, h1, h2, h3, out]"""
        out = [x]
        for L, act in zip(self.layers, self.acts):
            out.append(L.forward(out[-1], act=act))
        return out

    def backward(self, acts_: List[torch.Tensor], dout: torch.Tensor, need_dx: bool) -> Optional[torch.Tensor]:
        """Fills every layer's .grad; returns the gradient of the input rows when asked."""
        d = dout
        for i in range(len(self.layers) - 1, -1, -1):
            d = self.layers[i].backward(acts_[i], acts_[i + 1], d, need_dx=need_dx or i > 0, act=self.acts[i])
        return d

    def input_grad(self, acts_: List[torch.Tensor], dout: torch.Tensor) -> torch.Tensor:
        """Gradient of the input rows alone (no weight gradients)."""
        d = dout
        for i in range(len(self.layers) - 1, -1, -1):
            L, code = self.layers[i], DenseLayer._act_code(self.acts[i])
            if code:
                dz = torch.empty_like(d)
                check(lib.rlgan_act_backward(d.data_ptr(), acts_[i + 1].data_ptr(), dz.data_ptr(), d.numel(), code, _stream()),
                      "rlgan_act_backward")
                d = dz
            dx = torch.empty_like(acts_[i])
            sc, ns = L._slabs(d.shape[0], L.n, L.k)
            check(lib.rlgan_dense_backward_input_split(d.data_ptr(), L.n, L.canon.data_ptr(), dx.data_ptr(), L.k, d.shape[0], L.k, L.n,
                                                       None if sc is None else sc.data_ptr(), ns, _stream()),
                  "rlgan_dense_backward_input_split")
            d = dx
        return d


class _RowNet:
    """The four layers of one network as ONE flat parameter vector with the fragment images the row-pass kernels read
    (`DeviceNet`, kinds 4 / 5 of the library).  The layers' `canon` and `grad` become views into the flat vectors, so
    the layer-wise calls (acting, the tests' accessors, checkpoints) and the row passes see the same numbers."""

    def __init__(self, kind: int, a_pad: int, net: _Net, trainable: bool):
        flat = torch.cat([L.canon for L in net.layers]).cpu().numpy()
        self.dn = DeviceNet(kind, 256, a_pad, trainable=trainable, init=flat)
        self.grad = torch.zeros_like(self.dn.canon)
        o = 0
        for L in net.layers:
            L.canon, L.grad = self.dn.canon[o:o + L.n_params], self.grad[o:o + L.n_params]
            o += L.n_params
        assert o == self.dn.n_params
        self._ws = {}

    def workspace(self, rows: int) -> Workspace:
        if rows not in self._ws:
            self._ws[rows] = Workspace(self.dn, rows)
        return self._ws[rows]

    def repack(self) -> None:
        """After the flat vector was written from outside (a checkpoint): refresh the fragment images."""
        self.dn.repack()


def row_passes_fit(h_dim: int, a_pad: int) -> bool:
    """The row-pass kernels are built for the encoders' 256 features and 4 .. 12 (padded) action columns."""
    return h_dim == 256 and 4 <= a_pad <= 12


class LSTMUnit:
    """`lstm_unit` (Conv_model.py:67-99): one step of tf.contrib.rnn.BasicLSTMCell(n_lstm_unit) on the state encoder's
    features, state (c, h) in and out.  With USE_LSTM False (Conv_model.py:667) the loop only carries this state along and
    stores it with every transition (Curious_GAN_mario.py:69,75); the cell's variables get no gradient.  Parameters in
    canonical order: kernel [features + units, 4 units] (gate order i, j, f, o), then the bias."""

    FORGET_BIAS = 1.0                                            # BasicLSTMCell default

    def __init__(self, state_encoder: ConvEncoder, n_lstm_unit: Optional[int] = None, rng: Optional[np.random.RandomState] = None,
                 kernel: Optional[np.ndarray] = None, bias: Optional[np.ndarray] = None):
        rng = rng or np.random.RandomState()
        self.state_encoder = state_encoder
        self.h_dim = state_encoder.features
        self.units = int(n_lstm_unit or self.h_dim)              # Conv_model.py:685: n_lstm_unit = phis.shape[1]
        k = self.h_dim + self.units
        if kernel is None:                                       # the scope's default initializer (glorot uniform), zero bias
            kernel, bias = _xavier(rng, k, 4 * self.units), np.zeros(4 * self.units)
        self.cell = DenseLayer(k, 4 * self.units, kernel=kernel, bias=np.asarray(bias).reshape(-1))

    def get_initial_state(self):
        """[c, h], each zeros [1, units] (Conv_model.py:75-77,98-99)."""
        return [np.zeros((1, self.units), np.float32), np.zeros((1, self.units), np.float32)]

    def step_batch(self, phis: torch.Tensor, c_in: torch.Tensor, h_in: torch.Tensor):
        """(c', h') for a batch of features and states, all on the device."""
        n = phis.shape[0]
        x = torch.empty((n, self.h_dim + self.units), dtype=torch.float32, device=phis.device)
        check(lib.rlgan_concat_cols(phis.data_ptr(), row_ld(phis), self.h_dim, h_in.data_ptr(), row_ld(h_in), self.units, None, 0, 0,
                                    x.data_ptr(), x.shape[1], n, 0, _stream()), "rlgan_concat_cols")
        gates = self.cell.forward(x, act=False)
        c_in = c_in.contiguous()
        c_out, h_out = torch.empty_like(c_in), torch.empty_like(c_in)
        check(lib.rlgan_lstm_gates(gates.data_ptr(), c_in.data_ptr(), n, self.units, self.FORGET_BIAS, c_out.data_ptr(), h_out.data_ptr(),
                                   _stream()), "rlgan_lstm_gates")
        return c_out, h_out

    def get_state(self, s, lstm_state):
        """[c', h'] as NumPy arrays [1, units] for a single state [h, w, c] (Conv_model.py:93-96)."""
        s = torch.as_tensor(np.asarray(s, dtype=np.float32)[None], device=device())
        c = torch.as_tensor(np.asarray(lstm_state[0], dtype=np.float32).reshape(1, -1), device=device())
        h = torch.as_tensor(np.asarray(lstm_state[1], dtype=np.float32).reshape(1, -1), device=device())
        c2, h2 = self.step_batch(self.state_encoder.forward(s), c, h)
        return [c2.cpu().numpy(), h2.cpu().numpy()]


ACTOR_KEYS = ("w1", "b1", "w2", "b2", "w3", "b3", "wa", "ba")
CRITIC_KEYS = ("w1_s", "w1_a", "b1", "w2", "b2", "w3", "b3", "wq", "bq")


class PixelActor:
    def __init__(self, action_dim: int, learning_rate: float, t_replace_iter: int, state_encoder: ConvEncoder,
                 rng: Optional[np.random.RandomState] = None, params: Optional[Dict[str, np.ndarray]] = None,
                 target_params: Optional[Dict[str, np.ndarray]] = None):
        rng = rng or np.random.RandomState()
        self.a_dim, self.lr, self.t_replace_iter = int(action_dim), learning_rate, int(t_replace_iter)
        self.h_dim, self.a_pad = state_encoder.features, (self.a_dim + 3) // 4 * 4
        self.state_encoder = state_encoder
        self.eval_net = self._build(rng, params)
        self.target_net = self._build(rng, target_params)
        self.t_replace_counter = torch.zeros(1, dtype=torch.int32, device=device())
        # learn() as row passes over 16-row tiles (one launch for the step, one for the weight gradients, one for the
        # optimizer) where the shapes fit; `row_passes = False` keeps the layer-by-layer products
        self.row_passes = row_passes_fit(self.h_dim, self.a_pad)
        if self.row_passes:
            self._rn = _RowNet(_lib.NET_PIXEL_ACTOR, self.a_pad, self.eval_net, True)
            self._rn_t = _RowNet(_lib.NET_PIXEL_ACTOR, self.a_pad, self.target_net, False)
            self._ms = torch.ones_like(self._rn.dn.canon)
        self._opt = [_RMSProp(e, t) for e, t in zip(self.eval_net.layers, self.target_net.layers)]
        if self.row_passes:                                      # the layers' slots are views of the flat one
            o = 0
            for opt in self._opt:
                opt.ms = self._ms[o:o + opt.layer.n_params]
                o += opt.layer.n_params
        self._acts: Optional[List[torch.Tensor]] = None
        self.LSTM_unit = LSTMUnit(state_encoder, rng=rng)        # Conv_model.py:685-686,693-694
        self.LSTM_unit_ = LSTMUnit(state_encoder, rng=rng)

    def _build(self, rng, P) -> _Net:
        h, ad = self.h_dim, self.a_dim
        if P is None:                                           # xavier_initializer, biases 0.001, none on `a` (Conv_model.py:713-739)
            P = dict(w1=_xavier(rng, h, N_L1), b1=np.full(N_L1, 0.001), w2=_xavier(rng, N_L1, N_L2), b2=np.full(N_L2, 0.001),
                     w3=_xavier(rng, N_L2, N_L3), b3=np.full(N_L3, 0.001), wa=_xavier(rng, N_L3, ad), ba=np.zeros(ad))
        k3, b3 = _pad(P["w3"], P["b3"], N_L2, L3_PAD)
        ka, ba = _pad(P["wa"], P["ba"], L3_PAD, self.a_pad)
        return _Net([DenseLayer(h, N_L1, kernel=P["w1"], bias=np.asarray(P["b1"]).reshape(-1)),
                     DenseLayer(N_L1, N_L2, kernel=P["w2"], bias=np.asarray(P["b2"]).reshape(-1)),
                     DenseLayer(N_L2, L3_PAD, kernel=k3, bias=b3), DenseLayer(L3_PAD, self.a_pad, kernel=ka, bias=ba)],
                    ("relu6", "relu6", "relu6", "tanh"))

    # ---- batched forms -------------------------------------------------------------------------------
    def actions(self, phis: torch.Tensor, target: bool = False) -> torch.Tensor:
        """[n, a_pad] actions of the eval (or target) net on features; columns from a_dim on are zero."""
        net = self.target_net if target else self.eval_net
        acts_ = net.forward(phis.contiguous())
        if not target:
            self._acts = acts_
        return acts_[-1]

    def act_batch(self, s) -> torch.Tensor:
        """[n, a_dim] raw actions for a batch of states."""
        return self.actions(self.state_encoder.forward(s))[:, :self.a_dim]

    # ---- reference API -------------------------------------------------------------------------------
    def choose_action(self, s, lstm_state=None, var: float = 0.0) -> np.ndarray:
        """One-hot of argmax(normal(a, var)) for a single state [h, w, c] (Conv_model.py:756-765); the noise comes
        from NumPy's global stream like the reference's.  lstm_state is accepted and unused (USE_LSTM False)."""
        s = torch.as_tensor(np.asarray(s, dtype=np.float32)[None], device=device())
        a = self.act_batch(s).cpu().numpy().astype(np.float64)
        a = np.random.normal(a, var)
        onehot = np.zeros(self.a_dim, dtype=np.int64)
        onehot[a.argmax(axis=1)] = 1
        return onehot

    choose_mario_action = choose_action                          # Conv_model.py:767-776: the same body

    def learn(self, s, critic: "PixelCritic", b_lstm_s=None, phis: Optional[torch.Tensor] = None) -> None:
        """One RMSProp(-lr) step along the policy gradient, then the periodic target copy (Conv_model.py:746-754).
        phis: features of `s` when the caller already holds them."""
        if phis is None:
            phis = self.state_encoder.forward(s)
        if self.row_passes and critic.row_passes:
            self._learn_rows(phis.contiguous(), critic)
        else:
            self._learn_layers(phis, critic)
        # e_params / t_params are collected by scope (Conv_model.py:708-709), so the periodic tf.assign list also copies
        # the eval net's LSTM cell into the target's (Conv_model.py:752-753) although no optimizer touches either
        e, t = self.LSTM_unit.cell, self.LSTM_unit_.cell
        check(lib.rlgan_target_update(t.canon.data_ptr(), e.canon.data_ptr(), e.n_params, 1.0, self.t_replace_counter.data_ptr(),
                                      self.t_replace_iter, None, None, _stream()), "rlgan_target_update")

    def _learn_rows(self, phis: torch.Tensor, critic: "PixelCritic") -> None:
        n = phis.shape[0]
        ws = self._rn.workspace(rows16(n))
        step = _lib.PixelActorStep(self.a_pad, n, self._rn.dn.net(True), critic._rn.dn.net(True), mat(phis), ws.acts(),
                                   self.t_replace_counter.data_ptr())
        check(lib.rlgan_pixel_actor_step_rows(C.byref(step), _stream()), "rlgan_pixel_actor_step_rows")
        launch_wgrads([wgrad_job(self._rn.dn, ws)], 256, self.a_pad)
        red = reduce_job(self._rn.dn, ws)
        red.grad = self._rn.grad.data_ptr()
        launch_apply([rmsprop_job(self._rn.dn, self._rn.grad, self._ms, -self.lr, target=self._rn_t.dn, counter=self.t_replace_counter,
                                  period=self.t_replace_iter)], reduces=[red])

    def _learn_layers(self, phis: torch.Tensor, critic: "PixelCritic") -> None:
        a = self.actions(phis)
        cacts = critic.eval_net.forward(critic._rows(phis, a))
        dq = torch.ones_like(cacts[-1])
        dq[:, 1:] = 0                                            # only the first column is q
        dx = critic.eval_net.input_grad(cacts, dq)
        da = torch.empty_like(a)
        check(lib.rlgan_copy_cols(dx.data_ptr() + 4 * self.h_dim, critic.k_pad, da.data_ptr(), self.a_pad, a.shape[0], self.a_pad, 0,
                                  _stream()), "rlgan_copy_cols")
        self.eval_net.backward(self._acts, da, need_dx=False)
        self.t_replace_counter.add_(1)
        launch_apply([o.job(-self.lr, self.t_replace_counter, self.t_replace_iter) for o in self._opt])
        if self.row_passes:                                      # the row-pass kernels read the fragment images
            self._rn.repack()
            self._rn_t.repack()


class PixelCritic:
    def __init__(self, action_dim: int, learning_rate: float, gamma: float, t_replace_iter: int, actor: PixelActor,
                 state_encoder: ConvEncoder, rng: Optional[np.random.RandomState] = None,
                 params: Optional[Dict[str, np.ndarray]] = None, target_params: Optional[Dict[str, np.ndarray]] = None):
        rng = rng or np.random.RandomState()
        self.a_dim, self.lr, self.gamma, self.t_replace_iter = int(action_dim), learning_rate, float(gamma), int(t_replace_iter)
        self.actor, self.state_encoder = actor, state_encoder
        self.h_dim, self.a_pad = state_encoder.features, actor.a_pad
        self.k_pad = self.h_dim + self.a_pad
        self.eval_net = self._build(rng, params)
        self.target_net = self._build(rng, target_params)
        self.t_replace_counter = torch.zeros(1, dtype=torch.int32, device=device())
        self.loss_out = torch.zeros(1, dtype=torch.float32, device=device())
        self.row_passes = row_passes_fit(self.h_dim, self.a_pad) and actor.row_passes
        if self.row_passes:
            self._rn = _RowNet(_lib.NET_PIXEL_CRITIC, self.a_pad, self.eval_net, True)
            self._rn_t = _RowNet(_lib.NET_PIXEL_CRITIC, self.a_pad, self.target_net, False)
            self._ms = torch.ones_like(self._rn.dn.canon)
        self._opt = [_RMSProp(e, t) for e, t in zip(self.eval_net.layers, self.target_net.layers)]
        if self.row_passes:
            o = 0
            for opt in self._opt:
                opt.ms = self._ms[o:o + opt.layer.n_params]
                o += opt.layer.n_params
        self._parts: Optional[torch.Tensor] = None

    def _build(self, rng, P) -> _Net:
        h, ad = self.h_dim, self.a_dim
        if P is None:                                           # xavier_initializer, biases 0.01 (Conv_model.py:623-650)
            P = dict(w1_s=_xavier(rng, h, N_L1), w1_a=_xavier(rng, ad, N_L1), b1=np.full(N_L1, 0.01), w2=_xavier(rng, N_L1, N_L2),
                     b2=np.full(N_L2, 0.01), w3=_xavier(rng, N_L2, N_L3), b3=np.full(N_L3, 0.01), wq=_xavier(rng, N_L3, 1), bq=np.full(1, 0.01))
        k1 = np.zeros((self.k_pad, N_L1))
        k1[:h], k1[h:h + ad] = P["w1_s"], P["w1_a"]
        k3, b3 = _pad(P["w3"], P["b3"], N_L2, L3_PAD)
        kq, bq = _pad(P["wq"], P["bq"], L3_PAD, 4)
        return _Net([DenseLayer(self.k_pad, N_L1, kernel=k1, bias=np.asarray(P["b1"]).reshape(-1)),
                     DenseLayer(N_L1, N_L2, kernel=P["w2"], bias=np.asarray(P["b2"]).reshape(-1)),
                     DenseLayer(N_L2, L3_PAD, kernel=k3, bias=b3), DenseLayer(L3_PAD, 4, kernel=kq, bias=bq)],
                    ("relu6", "relu6", "relu6", False))

    def _rows(self, phis: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
        """[n, k_pad] rows [phis, a, 0]; a: [n, >= a_dim], its first a_dim columns are used."""
        n = phis.shape[0]
        x = torch.empty((n, self.k_pad), dtype=torch.float32, device=phis.device)
        check(lib.rlgan_concat_cols(phis.data_ptr(), row_ld(phis), self.h_dim, a.data_ptr(), row_ld(a), self.a_dim, None, 0, 0,
                                    x.data_ptr(), self.k_pad, n, self.a_pad - self.a_dim, _stream()), "rlgan_concat_cols")
        return x

    def q_batch(self, phis: torch.Tensor, a: torch.Tensor, target: bool = False) -> torch.Tensor:
        """[n, 1] q values on features."""
        net = self.target_net if target else self.eval_net
        return net.forward(self._rows(phis.contiguous(), a.contiguous()))[-1][:, :1]

    def learn(self, s, a: torch.Tensor, r: torch.Tensor, s_, b_lstm_s=None, b_lstm_s_=None, phis: Optional[torch.Tensor] = None,
              phis_next: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One RMSProp step on the TD error, then the periodic target copy (Conv_model.py:653-663); returns the loss
        before the step (device scalar).  r: [n, 1].  phis / phis_next: features when the caller already holds them."""
        if phis_next is None:
            phis_next = self.state_encoder.forward(s_)
        if phis is None:
            phis = self.state_encoder.forward(s)
        n = phis.shape[0]
        if self.row_passes:
            return self._learn_rows(phis.contiguous(), a.contiguous(), r.contiguous(), phis_next.contiguous())
        a_next = self.actor.actions(phis_next, target=True)
        q_next = self.target_net.forward(self._rows(phis_next, a_next))[-1]
        acts_ = self.eval_net.forward(self._rows(phis, a.contiguous()))
        q = acts_[-1]
        dq = torch.empty_like(q)
        n_parts = (n + 255) // 256
        if self._parts is None or self._parts.numel() < n_parts:
            self._parts = torch.empty(n_parts, dtype=torch.float32, device=q.device)
        check(lib.rlgan_td_error(q.data_ptr(), q_next.data_ptr(), 4, r.contiguous().data_ptr(), self.gamma, n, self._parts.data_ptr(),
                                 dq.data_ptr(), _stream()), "rlgan_td_error")
        check(lib.rlgan_sum_parts(self._parts.data_ptr(), n_parts, 1, 1.0 / n, self.loss_out.data_ptr(), _stream()), "rlgan_sum_parts")
        self.eval_net.backward(acts_, dq, need_dx=False)
        self.t_replace_counter.add_(1)
        launch_apply([o.job(self.lr, self.t_replace_counter, self.t_replace_iter) for o in self._opt])
        return self.loss_out

    def _learn_rows(self, phis: torch.Tensor, a: torch.Tensor, r: torch.Tensor, phis_next: torch.Tensor) -> torch.Tensor:
        n = phis.shape[0]
        ws = self._rn.workspace(rows16(n))
        n_tiles = rows16(n) // 16
        if self._parts is None or self._parts.numel() < n_tiles:
            self._parts = torch.empty(n_tiles, dtype=torch.float32, device=phis.device)
        step = _lib.PixelCriticStep(self.a_pad, self.a_dim, n, self._rn.dn.net(True), self._rn_t.dn.net(), self.actor._rn_t.dn.net(),
                                    mat(phis), mat(phis_next), mat(a), r.data_ptr(), self.gamma, ws.acts(), self._parts.data_ptr(),
                                    self.t_replace_counter.data_ptr())
        check(lib.rlgan_pixel_critic_step_rows(C.byref(step), _stream()), "rlgan_pixel_critic_step_rows")
        launch_wgrads([wgrad_job(self._rn.dn, ws)], 256, self.a_pad)
        red = reduce_job(self._rn.dn, ws, self._parts, n_tiles, 1.0 / n, self.loss_out)
        red.grad = self._rn.grad.data_ptr()
        launch_apply([rmsprop_job(self._rn.dn, self._rn.grad, self._ms, self.lr, target=self._rn_t.dn, counter=self.t_replace_counter,
                                  period=self.t_replace_iter)], reduces=[red])
        return self.loss_out
