"""The inverse-model restatement (oracle/inverse_model.py) against torch autograd (batch_norm with batch
statistics, relu6, cross_entropy): TensorFlow is absent, so this is the independent check (Conv_model.py:787-881)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import inverse_model as om


def _params(rng, k, ac):
    u = lambda lim, shape: rng.uniform(-lim, lim, shape)
    return dict(w1_s=u(0.1, (k, 500)), w1_s_=u(0.1, (k, 500)), b1=np.full(500, 0.001), g1=u(1, 500) + 1.5, be1=u(0.5, 500),
                w2=u(0.1, (500, 100)), b2=np.full(100, 0.001), g2=u(1, 100) + 1.5, be2=u(0.5, 100),
                w3=u(0.2, (100, ac)), b3=np.full(ac, 0.001), g3=u(1, ac) + 1.5, be3=u(0.5, ac))


def test_net_against_torch_autograd():
    rng = np.random.RandomState(0)
    n, k, ac = 9, 32, 6
    P = _params(rng, k, ac)
    phis, phis_n = rng.standard_normal((n, k)), rng.standard_normal((n, k))
    a = rng.uniform(0, 1, (n, ac))
    logits, cache = om.net_forward(phis, phis_n, P, True)
    loss, dlogits = om.softmax_ce(logits, a)
    dphis, dphis_n, G = om.net_backward(dlogits, P, cache)
    T = {k_: torch.tensor(v, requires_grad=True) for k_, v in P.items()}
    x, xn = torch.tensor(phis, requires_grad=True), torch.tensor(phis_n, requires_grad=True)
    bn = lambda z, g, b: F.batch_norm(z, None, None, g, b, training=True, eps=om.EPS)
    h1 = F.relu6(bn(x @ T["w1_s"] + xn @ T["w1_s_"] + T["b1"], T["g1"], T["be1"]))
    h2 = F.relu6(bn(h1 @ T["w2"] + T["b2"], T["g2"], T["be2"]))
    lt = bn(h2 @ T["w3"] + T["b3"], T["g3"], T["be3"])
    lo = F.cross_entropy(lt, torch.tensor(np.argmax(a, 1)))
    lo.backward()
    np.testing.assert_allclose(logits, lt.detach().numpy(), rtol=1e-10, atol=1e-10)
    assert abs(loss - lo.item()) < 1e-12
    np.testing.assert_allclose(dphis, x.grad.numpy(), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(dphis_n, xn.grad.numpy(), rtol=1e-8, atol=1e-12)
    for k_ in P:
        np.testing.assert_allclose(G[k_], T[k_].grad.numpy(), rtol=1e-7, atol=1e-11, err_msg=k_)


def test_inference_uses_the_initial_moving_statistics():
    rng = np.random.RandomState(1)
    P = _params(rng, 8, 4)
    z = rng.standard_normal((5, 500))
    y, _ = om.bn_forward(z, P["g1"], P["be1"], training=False)
    np.testing.assert_allclose(y, P["g1"] * z / np.sqrt(1 + om.EPS) + P["be1"], rtol=1e-12, atol=1e-12)
