"""Inverse model of the reference's pixel variant (`StateActionPredictor`, Conv_model.py:787-881):

    z1 = phis w1_s + phis' w1_s_ + b1          500 units        Conv_model.py:831-836
    h1 = relu6(BN(z1));  h2 = relu6(BN(dense100(h1)));  a_logit = BN(dense(ac_space)(h2))     Conv_model.py:838-850
    loss = mean sparse softmax cross-entropy(a_logit, argmax(a))                               Conv_model.py:801,818
    train_op = Adam(lr).minimize(loss, var_list = A_pred + the state encoder's variables)      Conv_model.py:811-820

phis and phis' are the state encoder's features of s and s'.  The mode test at Conv_model.py:807 compares against a
misspelt name and never matches, so the encoder is always in the variable list.  The l2 term is commented out of the
loss (Conv_model.py:819).  Batch normalisation uses batch statistics in `learn` and the initial moving statistics
(mean 0, variance 1: the reference never runs the update ops) in `eval`.

s and s' go through the encoder as one batch of 2n frames (s rows first), so one backward pass leaves the sum of both
contributions in the encoder's gradient buffers.  The logit layer is kept a multiple of 4 wide; the extra units have
zero weights, gamma 1 and beta 0, and receive zero gradients."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from ._lib import check, lib
from .conv import ConvEncoder, DenseLayer, FrameRef
from .core import _stream, device, row_ld
from .pixel_gan import _adam_for, _adam_jobs

N_L1, N_L2 = 500, 100                                           # Conv_model.py:832,841
BN_EPS = 1e-3                                                   # tf.layers.batch_normalization default


class BatchNorm:
    """gamma and beta of one tf.layers.batch_normalization (canonical order: gamma, then beta) with the buffers of
    its latest training pass."""

    def __init__(self, n: int, gamma: Optional[np.ndarray] = None, beta: Optional[np.ndarray] = None):
        self.n, self.n_params = n, 2 * n
        g = np.ones(n) if gamma is None else np.asarray(gamma).reshape(-1)
        b = np.zeros(n) if beta is None else np.asarray(beta).reshape(-1)
        self.canon = torch.from_numpy(np.concatenate([g, b]).astype(np.float32)).to(device())
        self.grad = torch.zeros(2 * n, dtype=torch.float32, device=device())
        self.stats = torch.zeros(2 * n, dtype=torch.float32, device=device())      # batch mean, then 1 / sqrt(var + eps)
        self._scratch: Optional[torch.Tensor] = None

    def _scratch_ptr(self, m: int):
        """Per-chunk partial results of a pass over m rows (none for the one-launch form)."""
        chunks = lib.rlgan_batchnorm_chunks(m)
        if chunks <= 1:
            return None
        if self._scratch is None or self._scratch.numel() < 2 * chunks * self.n:
            self._scratch = torch.empty(2 * chunks * self.n, dtype=torch.float32, device=device())
        return self._scratch.data_ptr()

    def _ptrs(self, t: torch.Tensor):
        return t.data_ptr(), t.data_ptr() + 4 * self.n

    def forward(self, z: torch.Tensor, training: bool, relu6: bool) -> torch.Tensor:
        assert z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and z.shape[1] == self.n
        y = torch.empty_like(z)
        ga, be = self._ptrs(self.canon)
        mean, inv = self._ptrs(self.stats)
        check(lib.rlgan_batchnorm_forward(z.data_ptr(), self.n, ga, be, z.shape[0], self.n, BN_EPS, 1 if training else 0,
                                          3 if relu6 else 0, y.data_ptr(), self.n, mean, inv, self._scratch_ptr(z.shape[0]), _stream()),
              "rlgan_batchnorm_forward")
        return y

    def backward(self, dy: torch.Tensor, y: torch.Tensor, z: torch.Tensor, relu6: bool) -> torch.Tensor:
        """Gradient of the training-mode pass that produced y from z: fills self.grad, returns dz."""
        dz = torch.empty_like(z)
        ga, _ = self._ptrs(self.canon)
        mean, inv = self._ptrs(self.stats)
        dga, dbe = self._ptrs(self.grad)
        check(lib.rlgan_batchnorm_backward(dy.data_ptr(), self.n, y.data_ptr(), self.n, z.data_ptr(), self.n, ga, mean, inv, z.shape[0],
                                           self.n, 3 if relu6 else 0, dz.data_ptr(), self.n, dga, dbe, self._scratch_ptr(z.shape[0]), _stream()),
              "rlgan_batchnorm_backward")
        return dz


class StateActionPredictor:
    def __init__(self, ac_space: int, learning_rate: float, state_encoder: ConvEncoder,
                 rng: Optional[np.random.RandomState] = None, params: Optional[dict] = None):
        """params: w1_s, w1_s_ [h, 500], b1, g1, be1, w2 [500, 100], b2, g2, be2, w3 [100, ac_space], b3, g3, be3."""
        rng = rng or np.random.RandomState()
        self.ac, self.lr, self.h_dim = int(ac_space), learning_rate, state_encoder.features
        self.ac_pad = (self.ac + 3) // 4 * 4
        assert self.h_dim % 4 == 0
        h = self.h_dim
        if params is None:                                      # xavier_initializer, biases 0.001 (Conv_model.py:828-829)
            xav = lambda i, o: rng.uniform(-np.sqrt(6.0 / (i + o)), np.sqrt(6.0 / (i + o)), (i, o))
            params = dict(w1_s=xav(h, N_L1), w1_s_=xav(h, N_L1), b1=np.full(N_L1, 0.001), w2=xav(N_L1, N_L2), b2=np.full(N_L2, 0.001),
                          w3=xav(N_L2, self.ac), b3=np.full(self.ac, 0.001))
        k3, b3 = np.zeros((N_L2, self.ac_pad)), np.zeros(self.ac_pad)
        k3[:, :self.ac], b3[:self.ac] = params["w3"], np.asarray(params["b3"]).reshape(-1)
        g3, be3 = np.ones(self.ac_pad), np.zeros(self.ac_pad)
        if "g3" in params:
            g3[:self.ac], be3[:self.ac] = np.asarray(params["g3"]).reshape(-1), np.asarray(params["be3"]).reshape(-1)
        self.l1 = DenseLayer(2 * h, N_L1, kernel=np.concatenate([params["w1_s"], params["w1_s_"]], 0),
                             bias=np.asarray(params["b1"]).reshape(-1))
        self.l2 = DenseLayer(N_L1, N_L2, kernel=params["w2"], bias=np.asarray(params["b2"]).reshape(-1))
        self.l3 = DenseLayer(N_L2, self.ac_pad, kernel=k3, bias=b3)
        self.bn1 = BatchNorm(N_L1, params.get("g1"), params.get("be1"))
        self.bn2 = BatchNorm(N_L2, params.get("g2"), params.get("be2"))
        self.bn3 = BatchNorm(self.ac_pad, g3, be3)
        self.state_encoder = state_encoder
        self.step = torch.zeros(1, dtype=torch.int32, device=device())
        self.loss_out = torch.zeros(1, dtype=torch.float32, device=device())
        trained = [self.l1, self.bn1, self.l2, self.bn2, self.l3, self.bn3] + state_encoder.layers + [state_encoder.head]
        self._opt = [_adam_for(L) for L in trained]
        self._rows: Optional[torch.Tensor] = None

    # ---- pieces ------------------------------------------------------------------------------------
    def _features(self, s, s_next: Optional[torch.Tensor]) -> torch.Tensor:
        """[n, 2h] rows [phis, phis'] from one encoder pass over the 2n frames.  s: float32 frames [n, h, w, c] with
        s_next alike, or a FrameRef naming the 2n frames in the uint8 replay, s first (`FrameMemory.ref_pair`), with
        s_next None."""
        if isinstance(s, FrameRef):
            assert s_next is None and s.n % 2 == 0
            n = s.n // 2
            f = self.state_encoder.forward(s)
        else:
            n = s.shape[0]
            f = self.state_encoder.forward(torch.cat([s, s_next], 0).contiguous())
        x = torch.empty((n, 2 * self.h_dim), dtype=torch.float32, device=f.device)
        check(lib.rlgan_concat_cols(f.data_ptr(), self.h_dim, self.h_dim, f.data_ptr() + 4 * n * self.h_dim, self.h_dim, self.h_dim,
                                    None, 0, 0, x.data_ptr(), 2 * self.h_dim, n, 0, _stream()), "rlgan_concat_cols")
        return x

    def _net(self, x: torch.Tensor, training: bool):
        z1 = self.l1.forward(x, act=False)
        h1 = self.bn1.forward(z1, training, True)
        z2 = self.l2.forward(h1, act=False)
        h2 = self.bn2.forward(z2, training, True)
        z3 = self.l3.forward(h2, act=False)
        return z1, h1, z2, h2, z3, self.bn3.forward(z3, training, False)

    def _loss(self, logits: torch.Tensor, a: torch.Tensor, want_grad: bool) -> Optional[torch.Tensor]:
        m = logits.shape[0]
        if self._rows is None or self._rows.numel() < m:
            self._rows = torch.empty(m, dtype=torch.float32, device=logits.device)
        d = torch.empty_like(logits) if want_grad else None
        check(lib.rlgan_softmax_ce(logits.data_ptr(), self.ac_pad, self.ac, a.data_ptr(), row_ld(a), m, self._rows.data_ptr(),
                                   None if d is None else d.data_ptr(), _stream()), "rlgan_softmax_ce")
        check(lib.rlgan_sum_parts(self._rows.data_ptr(), m, 1, 1.0 / m, self.loss_out.data_ptr(), _stream()), "rlgan_sum_parts")
        return d

    def logits(self, b_s: torch.Tensor, b_s_: torch.Tensor, training: bool = False) -> torch.Tensor:
        """a_logit [n, ac_space] (Conv_model.py:803)."""
        return self._net(self._features(b_s, b_s_), training)[-1][:, :self.ac]

    # ---- reference API -----------------------------------------------------------------------------
    def eval(self, b_s: torch.Tensor, b_s_: torch.Tensor, b_a: torch.Tensor) -> float:
        """loss with A_pred_is_training False (Conv_model.py:869-881)."""
        out = self._net(self._features(b_s, b_s_), False)[-1]
        self._loss(out, b_a.contiguous(), False)
        return float(self.loss_out.item())

    def learn(self, b_s: torch.Tensor, b_s_: torch.Tensor, b_a: torch.Tensor, lr: Optional[float] = None) -> torch.Tensor:
        """One Adam step (Conv_model.py:864-867); returns the loss before the step (device scalar).  b_s may be a
        FrameRef over both frame fields (see `_features`), b_s_ then None."""
        x = self._features(b_s, b_s_)
        n = x.shape[0]
        z1, h1, z2, h2, z3, out = self._net(x, True)
        dout = self._loss(out, b_a.contiguous(), True)
        dz3 = self.bn3.backward(dout, out, z3, False)
        dh2 = self.l3.backward(h2, z3, dz3, act=False)
        dz2 = self.bn2.backward(dh2, h2, z2, True)
        dh1 = self.l2.backward(h1, z2, dz2, act=False)
        dz1 = self.bn1.backward(dh1, h1, z1, True)
        dx = self.l1.backward(x, z1, dz1, act=False)
        df = torch.empty((2 * n, self.h_dim), dtype=torch.float32, device=dx.device)
        for c in range(2):
            check(lib.rlgan_copy_cols(dx.data_ptr() + 4 * c * self.h_dim, 2 * self.h_dim, df.data_ptr() + 4 * c * n * self.h_dim, self.h_dim,
                                      n, self.h_dim, 0, _stream()), "rlgan_copy_cols")
        self.state_encoder.backward(df)
        self.step.add_(1)
        _adam_jobs(self._opt, self.step, self.lr if lr is None else lr)
        return self.loss_out
