"""Pixel encoder on the device against the float64 restatement: forward features, input gradient and
every kernel/bias gradient, for the doom head (Conv_model.py:15-22) on 84x84x4 frames and the universe
head (Conv_model.py:36-49) on 84x84x4 (two layers -> 20x20x32 = 12,800 features) and 42x42x4 (four
layers).  Tolerance 1e-5 relative to the tensor scale, fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n,hw,layers", [("doom", 5, 84, None), ("universe2", 3, 84, 2), ("universe4", 19, 42, 4)])
def test_encoder_forward_and_backward(name, n, hw, layers):
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    spec = oc.DOOM_HEAD if layers is None else oc.universe_head(layers)
    rng = np.random.RandomState(3)
    x = rng.uniform(0, 1, (n, hw, hw, 4)).astype(np.float32).astype(np.float64)
    params = [(w.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64))
              for w, b in oc.init_encoder(rng, spec, 4, bias=0.001)]
    y, caches = oc.encoder_forward(x, params, spec)
    dy = rng.standard_normal(y.shape).astype(np.float32).astype(np.float64)
    dx, grads = oc.encoder_backward(dy, params, spec, caches)

    enc = R.ConvEncoder(hw, hw, 4, spec, params=params)
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    feat = enc.forward(xt)
    assert feat.shape == (n, y[0].size) and enc.features == y[0].size
    close(feat.cpu().numpy(), y.reshape(n, -1), what=f"{name} features")
    dxt = enc.backward(torch.from_numpy(dy.reshape(n, -1).astype(np.float32)).cuda(), need_dx=True)
    close(dxt.cpu().numpy(), dx, what=f"{name} dx")
    for i, (L, (dw, db)) in enumerate(zip(enc.layers, grads)):
        g = L.grad.cpu().numpy()
        close(g[:dw.size].reshape(dw.shape), dw, what=f"{name} layer {i} dW")
        close(g[dw.size:], db, what=f"{name} layer {i} db")


def test_conv_argument_errors():
    import ctypes as C
    from rl_gan_amd import _lib
    g = _lib.ConvGeom()
    assert _lib.lib.rlgan_conv_geometry(_lib.Conv(84, 84, 3, 8, 5, 4), C.byref(g)) == -2      # c_in not a multiple of 4
    assert _lib.lib.rlgan_conv_geometry(_lib.Conv(84, 84, 4, 128, 3, 2), C.byref(g)) == -2    # more than four column tiles
    assert _lib.lib.rlgan_conv_geometry(_lib.Conv(84, 84, 4, 8, 5, 4), C.byref(g)) == 0
    assert (g.h_out, g.w_out, g.k_len, g.n_params) == (20, 20, 100, 808)
