"""Generator of the reference's pixel variant in its default 'full_prediction' mode
(`Conv_model.Generator`, Conv_model.py:343-376,437-475,522-541; wiring Curious_GAN_mario.py:328-353):

    code  = relu([phis, a] * [w1_s; w1_a] + b1)                     200 units          Conv_model.py:444-455
    image = decoder(code)                                           42 x 42 x 1        Conv_model.py:457-473
    fake_observation = single-frame encoder(image)                  256 features       Curious_GAN_mario.py:340-343
    G_loss = mean((fake_observation - observation)^2),  observation = the same encoder on the last frame of s'
    train_op = Adam(lr).minimize(G_loss, var_list = Generator variables)               Conv_model.py:527-531

The single-frame encoder belongs to the discriminator's variable list in this mode (Conv_model.py:192-194); the
generator's step differentiates through it and leaves it unchanged.  Batch norm off (Curious_GAN_mario.py:384).
The 1-channel image is padded to the 4 channels the convolution kernels take; the encoder's first kernel has
zero rows for the three padding channels, which zero inputs keep at zero."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .conv import ConvEncoder, DenseLayer
from .core import _stream, device, launch_apply
from .deconv import Decoder

N_L1 = 200                                                      # Conv_model.py:445


def single_frame_encoder(hw: int = 42, n_convs: int = 2, rng: Optional[np.random.RandomState] = None, params=None,
                         dense_params=None, bias_init: float = 0.0) -> ConvEncoder:
    """`universe_feature_abstraction(single_S_, time_step=1, nConvs=2)` (Curious_GAN_mario.py:313-314) on frames
    padded from 1 to 4 channels.  params: [(kernel [3,3,1,32], bias), (kernel [3,3,32,32], bias), ...]."""
    from .conv import universe_head
    rng = rng or np.random.RandomState()
    if params is None:
        params, c = [], 1
        for f, k, _ in universe_head(n_convs):
            lim = np.sqrt(6.0 / (k * k * c + k * k * f))
            params.append((rng.uniform(-lim, lim, (k, k, c, f)), np.full(f, bias_init)))
            c = f
    k0, b0 = params[0]
    padded = np.zeros(k0.shape[:2] + (4, k0.shape[3]), dtype=np.float64)
    padded[:, :, :1, :] = k0
    return ConvEncoder(hw, hw, 4, universe_head(n_convs), rng=rng, bias_init=0.001, params=[(padded, b0)] + list(params[1:]),
                       dense=256, dense_params=dense_params)


class _Adam:
    """tf.train.AdamOptimizer slots of one layer (Conv_model.py:531)."""

    def __init__(self, layer, map_f=None, image=None):
        self.layer, self.map_f, self.image = layer, map_f, image
        self.m = torch.zeros_like(layer.canon)
        self.v = torch.zeros_like(layer.canon)

    def job(self, step: torch.Tensor, lr: float) -> _lib.ApplyJob:
        L = self.layer
        return _lib.ApplyJob(1, L.n_params, L.canon.data_ptr(), L.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                             lr, 0.9, 0.999, 1e-8, step.data_ptr(), None if self.map_f is None else self.map_f.data_ptr(),
                             None if self.image is None else self.image.data_ptr(), None, None, None, None, 1.0, None, 0, None)


class PixelGenerator:
    def __init__(self, h_dim: int, a_dim: int, learning_rate: float, obs_encoder: ConvEncoder,
                 rng: Optional[np.random.RandomState] = None, w1_s=None, w1_a=None, b1=None, decoder_params=None):
        rng = rng or np.random.RandomState()
        self.h_dim, self.a_dim, self.lr = h_dim, a_dim, learning_rate
        self.k_pad = (h_dim + a_dim + 3) // 4 * 4
        if w1_s is None:                                        # xavier_initializer, biases 0.001 (Conv_model.py:439-448)
            lim_s, lim_a = np.sqrt(6.0 / (h_dim + N_L1)), np.sqrt(6.0 / (a_dim + N_L1))
            w1_s, w1_a = rng.uniform(-lim_s, lim_s, (h_dim, N_L1)), rng.uniform(-lim_a, lim_a, (a_dim, N_L1))
            b1 = np.full(N_L1, 0.001)
        kernel = np.zeros((self.k_pad, N_L1))
        kernel[:h_dim], kernel[h_dim:h_dim + a_dim] = w1_s, w1_a
        self.l1 = DenseLayer(self.k_pad, N_L1, kernel=kernel, bias=np.asarray(b1).reshape(-1))
        self.decoder = Decoder(N_L1, rng=rng, params=decoder_params)
        self.obs_encoder = obs_encoder
        self.step = torch.zeros(1, dtype=torch.int32, device=device())
        self._opt: List[_Adam] = [_Adam(self.l1)] + [_Adam(L, L.map_w, L.wmat) for L in self.decoder.layers]
        self._x = self._h1 = self._img = None
        self._parts: Optional[torch.Tensor] = None
        self.loss_out = torch.zeros(1, dtype=torch.float32, device=device())

    # ---- pieces ------------------------------------------------------------------------------------
    def _input(self, phis: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
        n = phis.shape[0]
        x = torch.empty((n, self.k_pad), dtype=torch.float32, device=phis.device)
        check(lib.rlgan_copy_cols(phis.data_ptr(), phis.stride(0), x.data_ptr(), self.k_pad, n, self.h_dim, 0, _stream()), "rlgan_copy_cols")
        check(lib.rlgan_copy_cols(a.data_ptr(), a.stride(0), x.data_ptr() + 4 * self.h_dim, self.k_pad, n, self.a_dim,
                                  self.k_pad - self.h_dim - self.a_dim, _stream()), "rlgan_copy_cols")
        return x

    def predict_batch(self, phis: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
        """[n, 42, 42, 1] predicted next frames (Conv_model.py:562-567)."""
        self._x = self._input(phis.contiguous(), a.contiguous())
        self._h1 = self.l1.forward(self._x, act="relu")
        self._img = self.decoder.forward(self._h1)
        return self._img

    def _pad4(self, img1: torch.Tensor) -> torch.Tensor:
        n, h, w, _ = img1.shape
        out = torch.empty((n, h, w, 4), dtype=torch.float32, device=img1.device)
        check(lib.rlgan_copy_cols(img1.data_ptr(), 1, out.data_ptr(), 4, n * h * w, 1, 3, _stream()), "rlgan_copy_cols")
        return out

    def features(self, frames1: torch.Tensor) -> torch.Tensor:
        """Single-frame encoder on [n, h, w, 1] frames (Curious_GAN_mario.py:313-314,340-343)."""
        return self.obs_encoder.forward(self._pad4(frames1.contiguous()))

    def _mse(self, fake: torch.Tensor, real: torch.Tensor, want_grad: bool):
        n = fake.numel()
        n_parts = (n + 4095) // 4096
        if self._parts is None or self._parts.numel() < n_parts:
            self._parts = torch.empty(n_parts, dtype=torch.float32, device=fake.device)
        d = torch.empty_like(fake) if want_grad else None
        check(lib.rlgan_mse(fake.data_ptr(), real.data_ptr(), n, self._parts.data_ptr(), None if d is None else d.data_ptr(), _stream()),
              "rlgan_mse")
        check(lib.rlgan_sum_parts(self._parts.data_ptr(), n_parts, 1, 1.0 / n, self.loss_out.data_ptr(), _stream()), "rlgan_sum_parts")
        return d

    # ---- reference API -----------------------------------------------------------------------------
    def eval(self, phis: torch.Tensor, a: torch.Tensor, single_s_next: torch.Tensor) -> float:
        """G_loss (Conv_model.py:543-551)."""
        obs = self.features(single_s_next).clone()
        self._mse(self.features(self.predict_batch(phis, a)), obs, False)
        return float(self.loss_out.item())

    def learn(self, phis: torch.Tensor, a: torch.Tensor, single_s_next: torch.Tensor, lr: Optional[float] = None) -> torch.Tensor:
        """One Adam step on the generator's variables (Conv_model.py:533-541); returns the loss before the step
        (device scalar).  The encoder's gradient buffers are scratch here: its variables are not in the var_list."""
        obs = self.features(single_s_next).clone()
        fake = self.features(self.predict_batch(phis, a))      # the encoder now holds the activations of the fake pass
        dfake = self._mse(fake, obs, True)
        dimg4 = self.obs_encoder.backward(dfake, need_dx=True)
        n, h, w, _ = dimg4.shape
        dimg = torch.empty((n, h, w, 1), dtype=torch.float32, device=dimg4.device)
        check(lib.rlgan_copy_cols(dimg4.data_ptr(), 4, dimg.data_ptr(), 1, n * h * w, 1, 0, _stream()), "rlgan_copy_cols")
        dcode = self.decoder.backward(dimg)
        self.l1.backward(self._x, self._h1, dcode.contiguous(), need_dx=False, act="relu")
        self.step.add_(1)
        jobs = [o.job(self.step, self.lr if lr is None else lr) for o in self._opt]
        for i in range(0, len(jobs), 4):
            launch_apply(jobs[i:i + 4])
        return self.loss_out
