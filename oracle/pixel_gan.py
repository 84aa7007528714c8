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
