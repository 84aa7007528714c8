"""CPU restatement of the reference's pixel feature extractors.  TEST INFRASTRUCTURE ONLY: imported
by tests/ (and nothing in the product).

Layers are `tf.layers.conv2d(x, filters, [k, k], [s, s], activation=tf.nn.elu)` with TensorFlow's
defaults -- NHWC, padding 'valid', HWIO kernels -- as stacked by
  * the "doom" head, Conv_model.py:15-24 and Conv_DDPG.py:83-86: 8@5x5/4, 16@3x3/2, 32@3x3/2, 64@3x3/2;
  * the "universe" head, Conv_model.py:36-56 (norm=False branch): n x 32@3x3/2.
Parity unpinned: the arithmetic is TensorFlow's, which is not installable here, and the reference
holds no stored outputs for it (its Conv_DDPG graph does not even build: Conv_DDPG.py:88 calls conv2d
without a kernel size).  The restatement is cross-checked against torch.nn.functional.conv2d in
tests/test_oracle_conv.py.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

DOOM_HEAD = ((8, 5, 4), (16, 3, 2), (32, 3, 2), (64, 3, 2))          # (filters, kernel, stride)


def universe_head(n_convs: int = 4):
    return tuple((32, 3, 2) for _ in range(n_convs))


def out_size(n: int, k: int, s: int) -> int:
    return (n - k) // s + 1


def patches(x: np.ndarray, k: int, s: int) -> np.ndarray:
    """x [B,H,W,C] -> [B,Ho,Wo,k*k*C], patch elements ordered (ky, kx, c) like a flattened HWIO kernel."""
    B, H, W, C = x.shape
    Ho, Wo = out_size(H, k, s), out_size(W, k, s)
    sb, sh, sw, sc = x.strides
    v = np.lib.stride_tricks.as_strided(x, (B, Ho, Wo, k, k, C), (sb, sh * s, sw * s, sh, sw, sc), writeable=False)
    return v.reshape(B, Ho, Wo, k * k * C)


def elu(z):
    return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))


def conv_forward(x, w, b, k, s, act=True):
    """w: HWIO [k,k,Cin,Cout]; returns (y, cache)."""
    P = patches(x, k, s)
    z = P @ w.reshape(-1, w.shape[-1]) + b
    y = elu(z) if act else z
    return y, (x.shape, P, y)


def conv_backward(dy, w, cache, k, s, act=True):
    """Returns (dx, dw, db)."""
    xshape, P, y = cache
    dz = dy * np.where(y > 0, 1.0, y + 1.0) if act else dy         # ELU' from the output
    Cout = w.shape[-1]
    dw = (P.reshape(-1, P.shape[-1]).T @ dz.reshape(-1, Cout)).reshape(w.shape)
    db = dz.reshape(-1, Cout).sum(0)
    dP = dz @ w.reshape(-1, Cout).T                                   # [B,Ho,Wo,k*k*Cin]
    B, H, W, C = xshape
    Ho, Wo = dz.shape[1:3]
    dx = np.zeros(xshape, dtype=dy.dtype)
    dP = dP.reshape(B, Ho, Wo, k, k, C)
    for ky in range(k):
        for kx in range(k):
            dx[:, ky:ky + s * Ho:s, kx:kx + s * Wo:s, :] += dP[:, :, :, ky, kx, :]
    return dx, dw, db


def glorot_uniform(rng: np.random.RandomState, k: int, cin: int, cout: int) -> np.ndarray:
    """tf.layers.conv2d's default kernel initializer: limit = sqrt(6 / (fan_in + fan_out)) with the
    receptive field counted in both fans."""
    lim = np.sqrt(6.0 / (k * k * cin + k * k * cout))
    return rng.uniform(-lim, lim, (k, k, cin, cout))


def init_encoder(rng, layers: Sequence[Tuple[int, int, int]], c_in: int, bias: float = 0.0):
    params, c = [], c_in
    for f, k, s in layers:
        params.append((glorot_uniform(rng, k, c, f), np.full(f, bias)))
        c = f
    return params


def encoder_forward(x, params, layers):
    caches = []
    for (w, b), (f, k, s) in zip(params, layers):
        x, cache = conv_forward(x, w, b, k, s)
        caches.append(cache)
    return x, caches


def encoder_backward(dy, params, layers, caches):
    grads: List = []
    for (w, b), (f, k, s), cache in zip(reversed(params), reversed(layers), reversed(caches)):
        dy, dw, db = conv_backward(dy, w, cache, k, s)
        grads.append((dw, db))
    return dy, grads[::-1]


def dense_forward(x, w, b, act=True):
    """The 256-unit head after the flattened features (`tf.layers.dense(s, 256, activation=tf.nn.elu)`
    Conv_model.py:24; `dense1` then `tf.nn.elu` Conv_model.py:52-56).  w: [K, N]; returns (y, cache)."""
    z = x @ w + b
    y = elu(z) if act else z
    return y, (x, y)


def dense_backward(dy, w, cache, act=True):
    """Returns (dx, dw, db)."""
    x, y = cache
    dz = dy * np.where(y > 0, 1.0, y + 1.0) if act else dy         # ELU' from the output
    return dz @ w.T, x.T @ dz, dz.sum(0)
