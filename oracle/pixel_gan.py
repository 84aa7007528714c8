"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the generator of the pixel variant in 'full_prediction' mode.

Restates, in float64 NumPy, `Generator._build_full_generator` + `model_loss` (Conv_model.py:437-475,522-531)
with the wiring of Curious_GAN_mario.py:313-314,340-353: code = relu([phis, a] W1 + b1), image = decoder(code),
G_loss = mean((E(image) - E(single_s'))^2) with E the single-frame encoder (two 3x3/2 ELU convolutions and a
256-unit ELU layer), gradients with respect to the generator's variables only.

Parity unpinned against TensorFlow (not installed; the reference holds no fixture for this graph): the pieces
are cross-checked against torch autograd in tests/test_oracle_conv.py and tests/test_oracle_deconv.py."""
import numpy as np

from . import conv as oc
from . import deconv as od


def encoder(x, enc_params, enc_dense, layers):
    y, caches = oc.encoder_forward(x, enc_params, layers)
    f, dcache = oc.dense_forward(y.reshape(y.shape[0], -1), enc_dense[0], enc_dense[1])
    return f, (caches, dcache, y.shape)


def encoder_input_grad(df, enc_params, enc_dense, layers, cache):
    caches, dcache, yshape = cache
    dflat, _, _ = oc.dense_backward(df, enc_dense[0], dcache)
    dx, _ = oc.encoder_backward(dflat.reshape(yshape), enc_params, layers, caches)
    return dx


def generator_loss_and_grads(phis, a, single_s_next, w1_s, w1_a, b1, dec_params, enc_params, enc_dense):
    """Returns (loss, image, {"w1_s", "w1_a", "b1", "decoder": [(dw, db), ...]})."""
    layers = oc.universe_head(len(enc_params))
    obs, _ = encoder(single_s_next, enc_params, enc_dense, layers)
    z1 = phis @ w1_s + a @ w1_a + b1                                  # Conv_model.py:452
    code = np.maximum(z1, 0)                                          # Conv_model.py:455
    image, dcaches = od.decoder_forward(code, dec_params)
    fake, ecache = encoder(image, enc_params, enc_dense, layers)
    diff = fake - obs
    loss = np.mean(diff * diff)                                       # Conv_model.py:527
    dimage = encoder_input_grad(2.0 * diff / diff.size, enc_params, enc_dense, layers, ecache)
    dcode, dgrads = od.decoder_backward(dimage, dec_params, dcaches)
    dz1 = dcode * (z1 > 0)
    return loss, image, {"w1_s": phis.T @ dz1, "w1_a": a.T @ dz1, "b1": dz1.sum(0), "decoder": dgrads}


def encoder_grads(df, enc_params, enc_dense, layers, cache):
    """Returns (dx, [(dw, db) per conv layer], (dw, db) of the 256-unit layer)."""
    caches, dcache, yshape = cache
    dflat, dwd, dbd = oc.dense_backward(df, enc_dense[0], dcache)
    dx, grads = oc.encoder_backward(dflat.reshape(yshape), enc_params, layers, caches)
    return dx, grads, (dwd, dbd)


def sigmoid_ce(x, z):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x, 0) - x z + log(1 + exp(-|x|))."""
    return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))


def discriminator_net(h, g, a, P):
    """Conv_model.py:225-268 without batch norm: relu(h w1_s + g w1_G + a w1_a + b1) -> relu(dense 10) -> dense 1."""
    z1 = h @ P["w1_s"] + g @ P["w1_G"] + a @ P["w1_a"] + P["b1"]
    h1 = np.maximum(z1, 0)
    z3 = h1 @ P["w3"] + P["b3"]
    h3 = np.maximum(z3, 0)
    return h3 @ P["w2"] + P["b2"], (h, g, a, z1, h1, z3, h3)


def discriminator_net_backward(dx, P, cache):
    h, g, a, z1, h1, z3, h3 = cache
    G = {"w2": h3.T @ dx, "b2": dx.sum(0)}
    dz3 = (dx @ P["w2"].T) * (z3 > 0)
    G["w3"], G["b3"] = h1.T @ dz3, dz3.sum(0)
    dz1 = (dz3 @ P["w3"].T) * (z1 > 0)
    G.update(w1_s=h.T @ dz1, w1_G=g.T @ dz1, w1_a=a.T @ dz1, b1=dz1.sum(0))
    return dz1 @ P["w1_s"].T, dz1 @ P["w1_G"].T, G


def discriminator_loss_and_grads(s, a, single_s_next, g_image, P, st_params, st_dense, enc_params, enc_dense):
    """'full_prediction' discriminator step (Conv_model.py:183-223, Curious_GAN_mario.py:340-346): the generated
    image is fed as data, so nothing flows to the generator; the variables are the discriminator's, the state
    encoder's and the single-frame encoder's.  Returns (loss_real, loss_fake, grads)."""
    lay_s, lay_o = oc.universe_head(len(st_params)), oc.universe_head(len(enc_params))
    phis, scache = encoder(s, st_params, st_dense, lay_s)
    obs, rcache = encoder(single_s_next, enc_params, enc_dense, lay_o)
    fobs, fcache = encoder(g_image, enc_params, enc_dense, lay_o)
    xr, cr = discriminator_net(phis, obs, a, P)
    xf, cf = discriminator_net(phis, fobs, a, P)
    n = s.shape[0]
    loss_real, loss_fake = sigmoid_ce(xr, 1.0).mean(), sigmoid_ce(xf, 0.0).mean()
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    dh_r, dg_r, Gr = discriminator_net_backward((sig(xr) - 1.0) / n, P, cr)
    dh_f, dg_f, Gf = discriminator_net_backward(sig(xf) / n, P, cf)
    grads = {k: Gr[k] + Gf[k] for k in Gr}
    _, gs, gsd = encoder_grads(dh_r + dh_f, st_params, st_dense, lay_s, scache)
    _, g1, g1d = encoder_grads(dg_r, enc_params, enc_dense, lay_o, rcache)
    _, g2, g2d = encoder_grads(dg_f, enc_params, enc_dense, lay_o, fcache)
    grads["state_encoder"] = gs + [gsd]
    grads["obs_encoder"] = [(x[0] + y[0], x[1] + y[1]) for x, y in zip(g1 + [g1d], g2 + [g2d])]
    return loss_real, loss_fake, grads
