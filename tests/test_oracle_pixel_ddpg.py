"""The pixel actor / critic restatement (oracle/pixel_ddpg.py) against torch autograd: TensorFlow is absent, so this is
the independent check (Conv_model.py:569-784)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import pixel_ddpg as od


def params(rng, h, ad, spread=1.0):
    u = lambda lim, shape: rng.uniform(-lim, lim, shape) * spread
    A = dict(w1=u(0.15, (h, 200)), b1=u(0.3, 200), w2=u(0.15, (200, 200)), b2=u(0.3, 200), w3=u(0.3, (200, 10)), b3=u(0.3, 10) + 0.5,
             wa=u(0.6, (10, ad)), ba=u(0.1, ad))
    C = dict(w1_s=u(0.15, (h, 200)), w1_a=u(0.3, (ad, 200)), b1=u(0.3, 200), w2=u(0.15, (200, 200)), b2=u(0.3, 200),
             w3=u(0.3, (200, 10)), b3=u(0.3, 10) + 0.5, wq=u(0.6, (10, 1)), bq=u(0.1, 1))
    return A, C


def _t(P):
    return {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}


def _actor(x, A):
    h = F.relu6(x @ A["w1"] + A["b1"]); h = F.relu6(h @ A["w2"] + A["b2"]); h = F.relu6(h @ A["w3"] + A["b3"])
    return torch.tanh(h @ A["wa"] + A["ba"])


def _critic(x, a, C):
    h = F.relu6(x @ C["w1_s"] + a @ C["w1_a"] + C["b1"]); h = F.relu6(h @ C["w2"] + C["b2"]); h = F.relu6(h @ C["w3"] + C["b3"])
    return h @ C["wq"] + C["bq"]


def test_critic_step_against_torch_autograd():
    rng = np.random.RandomState(0)
    n, h, ad = 11, 24, 6
    A, C = params(rng, h, ad)
    At, Ct = params(rng, h, ad)
    phis, phis_n = rng.uniform(0, 3, (n, h)), rng.uniform(0, 3, (n, h))
    a, r = np.eye(ad)[rng.randint(0, ad, n)], rng.uniform(0, 1, (n, 1))
    loss, G = od.critic_loss_and_grads(phis, a, r, phis_n, C, Ct, At, 0.99)
    Ce = _t(C)
    with torch.no_grad():
        y = torch.tensor(r) + 0.99 * _critic(torch.tensor(phis_n), _actor(torch.tensor(phis_n), _t(At)), _t(Ct))
    lo = ((y - _critic(torch.tensor(phis), torch.tensor(a), Ce)) ** 2).mean()
    lo.backward()
    assert abs(loss - lo.item()) < 1e-12
    for k in C:
        np.testing.assert_allclose(G[k], Ce[k].grad.numpy().reshape(G[k].shape), rtol=1e-9, atol=1e-13, err_msg=k)


def test_policy_gradient_against_torch_autograd():
    rng = np.random.RandomState(1)
    n, h, ad = 9, 24, 6
    A, C = params(rng, h, ad)
    phis = rng.uniform(0, 3, (n, h))
    a, da, G = od.actor_policy_grads(phis, A, C)
    Ae = _t(A)
    at = _actor(torch.tensor(phis), Ae)
    _critic(torch.tensor(phis), at, _t(C)).sum().backward()          # d(sum q)/d(theta) = sum_b da/dtheta * dq/da
    np.testing.assert_allclose(a, at.detach().numpy(), rtol=1e-12, atol=1e-12)
    for k in A:
        np.testing.assert_allclose(G[k], Ae[k].grad.numpy().reshape(G[k].shape), rtol=1e-9, atol=1e-13, err_msg=k)
    assert np.abs(da).max() > 0 and (np.abs(od.actor_forward(phis, A)[1][5]) > 0).any()


def test_lstm_step_against_torch_cell():
    """torch.nn.LSTMCell orders its gates i, f, g, o and has no forget bias: TensorFlow's kernel is re-ordered and the
    forget bias folded into the bias to compare."""
    rng = np.random.RandomState(2)
    n, k, u = 5, 12, 8
    kernel, bias = rng.uniform(-0.4, 0.4, (k + u, 4 * u)), rng.uniform(-0.2, 0.2, 4 * u)
    x, c, h = rng.standard_normal((n, k)), rng.standard_normal((n, u)), rng.uniform(-1, 1, (n, u))
    c2, h2 = od.lstm_step(x, c, h, kernel, bias)
    i, j, f, o = np.split(kernel, 4, axis=1)
    bi, bj, bf, bo = np.split(bias, 4)
    cell = torch.nn.LSTMCell(k, u).double()
    W = np.concatenate([i, f, j, o], 1)                              # torch: i, f, g (= j), o
    with torch.no_grad():
        cell.weight_ih.copy_(torch.tensor(W[:k].T)); cell.weight_hh.copy_(torch.tensor(W[k:].T))
        cell.bias_ih.copy_(torch.tensor(np.concatenate([bi, bf + 1.0, bj, bo]))); cell.bias_hh.zero_()
        ht, ct = cell(torch.tensor(x), (torch.tensor(h), torch.tensor(c)))
    np.testing.assert_allclose(c2, ct.numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(h2, ht.numpy(), rtol=1e-12, atol=1e-12)
