"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the inverse model of the pixel variant.

Restates `StateActionPredictor` (Conv_model.py:787-881) in float64 NumPy: logits = BN(dense(relu6(BN(dense(
relu6(BN(phis w1_s + phis' w1_s_ + b1))))))) with tf.layers.batch_normalization defaults (epsilon 1e-3, batch
statistics with the biased variance while training; the moving statistics stay at their initial 0 and 1 because
the reference never runs the update ops, so inference divides by sqrt(1 + eps) only), loss = mean sparse softmax
cross-entropy against argmax(a) (Conv_model.py:801,818); phis and phis' are the state encoder's features of s
and s', and the encoder is trained by this step too (the mode test at Conv_model.py:807 never matches).

Parity unpinned against TensorFlow (not installed; no fixture in the reference): checked against torch's
batch_norm / cross_entropy autograd and finite differences in tests/test_oracle_inverse_model.py."""
import numpy as np

from . import conv as oc
from . import pixel_gan as op

EPS = 1e-3


def bn_forward(z, gamma, beta, training=True):
    if training:
        mean, var = z.mean(0), z.var(0)
    else:
        mean, var = np.zeros(z.shape[1]), np.ones(z.shape[1])
    inv = 1.0 / np.sqrt(var + EPS)
    xh = (z - mean) * inv
    return gamma * xh + beta, (xh, inv, gamma)


def bn_backward(dy, cache):
    xh, inv, gamma = cache
    m = dy.shape[0]
    dg, db = (dy * xh).sum(0), dy.sum(0)
    return gamma * inv * (dy - db / m - xh * dg / m), dg, db


def relu6(x):
    return np.clip(x, 0, 6)


def net_forward(phis, phis_next, P, training=True):
    z1 = phis @ P["w1_s"] + phis_next @ P["w1_s_"] + P["b1"]
    y1, c1 = bn_forward(z1, P["g1"], P["be1"], training)
    h1 = relu6(y1)
    z2 = h1 @ P["w2"] + P["b2"]
    y2, c2 = bn_forward(z2, P["g2"], P["be2"], training)
    h2 = relu6(y2)
    z3 = h2 @ P["w3"] + P["b3"]
    logits, c3 = bn_forward(z3, P["g3"], P["be3"], training)
    return logits, (phis, phis_next, y1, h1, y2, h2, c1, c2, c3)


def softmax_ce(logits, a):
    label = np.argmax(a, 1)
    x = logits - logits.max(1, keepdims=True)
    lse = np.log(np.exp(x).sum(1))
    rows = lse - x[np.arange(len(label)), label]
    p = np.exp(x - lse[:, None])
    p[np.arange(len(label)), label] -= 1.0
    return rows.mean(), p / len(label)


def net_backward(dlogits, P, cache):
    phis, phis_next, y1, h1, y2, h2, c1, c2, c3 = cache
    G = {}
    dz3, G["g3"], G["be3"] = bn_backward(dlogits, c3)
    G["w3"], G["b3"] = h2.T @ dz3, dz3.sum(0)
    dy2 = (dz3 @ P["w3"].T) * ((y2 > 0) & (y2 < 6))
    dz2, G["g2"], G["be2"] = bn_backward(dy2, c2)
    G["w2"], G["b2"] = h1.T @ dz2, dz2.sum(0)
    dy1 = (dz2 @ P["w2"].T) * ((y1 > 0) & (y1 < 6))
    dz1, G["g1"], G["be1"] = bn_backward(dy1, c1)
    G["w1_s"], G["w1_s_"], G["b1"] = phis.T @ dz1, phis_next.T @ dz1, dz1.sum(0)
    return dz1 @ P["w1_s"].T, dz1 @ P["w1_s_"].T, G


def loss_and_grads(s, s_next, a, P, st_params, st_dense):
    """Returns (loss, grads) with grads["state_encoder"] = [(dw, db) per conv layer, (dw, db) of the 256-unit layer]."""
    layers = oc.universe_head(len(st_params))
    phis, c_s = op.encoder(s, st_params, st_dense, layers)
    phis_next, c_n = op.encoder(s_next, st_params, st_dense, layers)
    logits, cache = net_forward(phis, phis_next, P, True)
    loss, dlogits = softmax_ce(logits, a)
    dphis, dphis_next, G = net_backward(dlogits, P, cache)
    _, g1, g1d = op.encoder_grads(dphis, st_params, st_dense, layers, c_s)
    _, g2, g2d = op.encoder_grads(dphis_next, st_params, st_dense, layers, c_n)
    G["state_encoder"] = [(x[0] + y[0], x[1] + y[1]) for x, y in zip(g1 + [g1d], g2 + [g2d])]
    return loss, G


def eval_loss(s, s_next, a, P, st_params, st_dense):
    layers = oc.universe_head(len(st_params))
    phis, _ = op.encoder(s, st_params, st_dense, layers)
    phis_next, _ = op.encoder(s_next, st_params, st_dense, layers)
    logits, _ = net_forward(phis, phis_next, P, False)
    return softmax_ce(logits, a)[0]
