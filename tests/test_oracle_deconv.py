"""The transposed-convolution restatement (oracle/deconv.py) against torch's conv_transpose2d and autograd:
TensorFlow is absent, so this is the independent check of the arithmetic (Conv_model.py:457-473)."""
import numpy as np
import pytest
import torch

from oracle import deconv as od


@pytest.mark.parametrize("h,ci,co,k,s,act", [(1, 8, 12, 3, 2, True), (3, 12, 8, 4, 2, True), (5, 8, 1, 6, 2, False), (4, 4, 4, 2, 3, True)])
def test_layer_against_torch(h, ci, co, k, s, act):
    rng = np.random.RandomState(h + k)
    x = rng.standard_normal((3, h, h, ci))
    w = rng.standard_normal((k, k, co, ci)) * 0.3
    b = rng.standard_normal(co) * 0.1
    y, cache = od.deconv_forward(x, w, b, s, act)
    dy = rng.standard_normal(y.shape)
    dx, dw, db = od.deconv_backward(dy, w, cache, s, act)
    xt = torch.tensor(x.transpose(0, 3, 1, 2), requires_grad=True)
    wt = torch.tensor(w.transpose(3, 2, 0, 1).copy(), requires_grad=True)       # torch: [in, out, kh, kw]
    bt = torch.tensor(b, requires_grad=True)
    yt = torch.nn.functional.conv_transpose2d(xt, wt, bt, stride=s)
    if act:
        yt = torch.relu(yt)
    assert yt.shape[2] == od.out_size(h, k, s)
    np.testing.assert_allclose(y, yt.detach().numpy().transpose(0, 2, 3, 1), rtol=1e-12, atol=1e-12)
    yt.backward(torch.tensor(dy.transpose(0, 3, 1, 2).copy()))
    np.testing.assert_allclose(dx, xt.grad.numpy().transpose(0, 2, 3, 1), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(dw, wt.grad.numpy().transpose(2, 3, 1, 0), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-11, atol=1e-11)


def test_decoder_shapes_follow_the_reference():
    rng = np.random.RandomState(0)
    params = od.init_decoder(rng, 200, bias=0.001)
    y, caches = od.decoder_forward(rng.standard_normal((2, 200)), params)
    assert [c[1].shape[1:] for c in caches] == [(3, 3, 128), (8, 8, 64), (19, 19, 32), (42, 42, 1)]   # Conv_model.py:457-473
    assert (caches[0][1] >= 0).all() and (caches[2][1] >= 0).all() and (y < 0).any()                  # ReLU between, none at the end
    dx, grads = od.decoder_backward(rng.standard_normal(y.shape), params, caches)
    assert dx.shape == (2, 200) and [g[0].shape for g in grads] == [p[0].shape for p in params]
