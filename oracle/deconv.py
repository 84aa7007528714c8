"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): transposed convolutions of the pixel generator's decoder.

Restates `tf.layers.conv2d_transpose(net, filters, k, strides=2)` (NHWC, padding 'valid', kernel
[k, k, out, in], bias) and the decoder stack of the reference's full-prediction generator,
1x1x200 -> 3x3x128 -> 8x8x64 -> 19x19x32 -> 42x42x1 with ReLU between the layers and none at the end
(Conv_model.py:457-473), forward and backward, in float64 NumPy.

Parity unpinned against TensorFlow (not installed; the reference holds no fixture for these layers): the
restatement is cross-checked against torch's conv_transpose2d and autograd in tests/test_oracle_deconv.py."""
from typing import Sequence, Tuple

import numpy as np

DECODER = ((128, 3, 2), (64, 4, 2), (32, 5, 2), (1, 6, 2))      # (filters, kernel, stride), Conv_model.py:458-473


def out_size(n: int, k: int, s: int) -> int:
    return (n - 1) * s + k


def deconv_forward(x, w, b, s, act=True):
    """x [n,h,w,ci]; w [k,k,co,ci] (TensorFlow's conv2d_transpose kernel); returns (y, cache)."""
    n, h, wd, _ = x.shape
    k, co = w.shape[0], w.shape[2]
    z = np.zeros((n, out_size(h, k, s), out_size(wd, k, s), co), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            z[:, ky:ky + s * h:s, kx:kx + s * wd:s, :] += x @ w[ky, kx].T
    z = z + b
    y = np.maximum(z, 0) if act else z
    return y, (x, y)


def deconv_backward(dy, w, cache, s, act=True):
    """Returns (dx, dw, db)."""
    x, y = cache
    n, h, wd, _ = x.shape
    k = w.shape[0]
    dz = dy * (y > 0) if act else dy
    dx, dw = np.zeros_like(x), np.zeros_like(w)
    for ky in range(k):
        for kx in range(k):
            g = dz[:, ky:ky + s * h:s, kx:kx + s * wd:s, :]
            dx += g @ w[ky, kx]
            dw[ky, kx] = np.einsum("nhwo,nhwi->oi", g, x)
    return dx, dw, dz.sum((0, 1, 2))


def init_decoder(rng: np.random.RandomState, c_in: int, layers: Sequence[Tuple[int, int, int]] = DECODER, bias: float = 0.0):
    """Glorot-uniform kernels (tf.layers default) and constant biases."""
    params = []
    for f, k, _ in layers:
        lim = np.sqrt(6.0 / (k * k * c_in + k * k * f))
        params.append((rng.uniform(-lim, lim, (k, k, f, c_in)), np.full(f, bias)))
        c_in = f
    return params


def decoder_forward(x, params, layers=DECODER):
    """x [n, c] -> image [n, H, W, filters of the last layer]; ReLU after every layer but the last."""
    x = x.reshape(x.shape[0], 1, 1, -1)
    caches = []
    for i, ((w, b), (_, _, s)) in enumerate(zip(params, layers)):
        x, c = deconv_forward(x, w, b, s, act=i + 1 < len(layers))
        caches.append(c)
    return x, caches


def decoder_backward(dy, params, caches, layers=DECODER):
    grads = []
    for i in range(len(layers) - 1, -1, -1):
        dy, dw, db = deconv_backward(dy, params[i][0], caches[i], layers[i][2], act=i + 1 < len(layers))
        grads.append((dw, db))
    return dy.reshape(dy.shape[0], -1), grads[::-1]
