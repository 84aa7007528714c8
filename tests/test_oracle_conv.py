"""The NumPy conv/ELU restatement against torch.nn.functional.conv2d + autograd (float64, CPU) for both
feature extractors of the reference (Conv_model.py:15-24,36-56)."""
import numpy as np
import pytest
import torch

from oracle import conv as oc


@pytest.mark.parametrize("layers,shape", [(oc.DOOM_HEAD, (3, 84, 84, 4)), (oc.universe_head(2), (2, 84, 84, 4)),
                                          (oc.universe_head(4), (2, 42, 42, 4))])
def test_encoder_matches_torch_autograd(layers, shape):
    rng = np.random.RandomState(0)
    x = rng.uniform(0, 1, shape)
    params = oc.init_encoder(rng, layers, shape[-1], bias=0.001)
    y, caches = oc.encoder_forward(x, params, layers)
    dy = rng.standard_normal(y.shape)
    dx, grads = oc.encoder_backward(dy, params, layers, caches)

    xt = torch.tensor(x.transpose(0, 3, 1, 2), requires_grad=True)
    tp = [(torch.tensor(w.transpose(3, 2, 0, 1).copy(), requires_grad=True), torch.tensor(b, requires_grad=True)) for w, b in params]
    h = xt
    for (w, b), (f, k, s) in zip(tp, layers):
        h = torch.nn.functional.elu(torch.nn.functional.conv2d(h, w, b, stride=s))
    np.testing.assert_allclose(y, h.detach().numpy().transpose(0, 2, 3, 1), rtol=1e-12, atol=1e-13)
    h.backward(torch.tensor(dy.transpose(0, 3, 1, 2).copy()))
    np.testing.assert_allclose(dx, xt.grad.numpy().transpose(0, 2, 3, 1), rtol=1e-10, atol=1e-12)
    for (dw, db), (w, b) in zip(grads, tp):
        np.testing.assert_allclose(dw, w.grad.numpy().transpose(2, 3, 1, 0), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(db, b.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_output_shapes_of_the_reference_heads():
    # 84x84 -> 20 -> 9 -> 4 -> 1 (64 features); universe head at 42x42: 20 -> 9 -> 4 -> 2 (Curious_GAN_mario.py:370-371)
    assert [oc.out_size(84, 5, 4), oc.out_size(20, 3, 2), oc.out_size(9, 3, 2), oc.out_size(4, 3, 2)] == [20, 9, 4, 1]
    assert [oc.out_size(84, 3, 2), oc.out_size(41, 3, 2)] == [41, 20]


def test_dense_head_matches_torch_autograd():
    """fc(256) + ELU (Conv_model.py:24): the restatement's three gradients against autograd."""
    import torch
    from oracle import conv as oc
    rng = np.random.RandomState(5)
    x, w, b = rng.standard_normal((9, 64)), rng.uniform(-0.2, 0.2, (64, 256)), rng.uniform(-0.1, 0.1, 256)
    y, cache = oc.dense_forward(x, w, b)
    dy = rng.standard_normal(y.shape)
    dx, dw, db = oc.dense_backward(dy, w, cache)
    xt, wt, bt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, w, b))
    yt = torch.nn.functional.elu(xt @ wt + bt)
    yt.backward(torch.tensor(dy))
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-12, atol=1e-12)
    for got, ref in ((dx, xt.grad), (dw, wt.grad), (db, bt.grad)):
        np.testing.assert_allclose(got, ref.numpy(), rtol=1e-10, atol=1e-12)
