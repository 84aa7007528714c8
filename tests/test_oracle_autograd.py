"""Independent derivation of the oracle's hand-written gradients with torch.autograd (CPU,
float64).  The reference's TensorFlow graph cannot be executed here, so this is the second,
independently derived opinion on every gradient tensor of the four update steps
(GAN-RL/DDPG.py:141-151,107-113; GAN-RL/Curious_GAN.py:33-45,137-142)."""
import numpy as np
import pytest
import torch

from oracle import nets
from oracle.agent import CuriousAgent, Hyper
from oracle.optim import Adam, RMSProp


def tparams(p):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}


def lrelu(x):
    return torch.maximum(0.2 * x, x)


def relu6(x):
    return torch.clamp(x, 0, 6)


def t_actor(p, s, hidden_only=False):
    h = relu6(s @ p["l1/kernel"] + p["l1/bias"])
    h = relu6(h @ p["l2/kernel"] + p["l2/bias"])
    h = torch.relu(h @ p["l3/kernel"] + p["l3/bias"])
    z = h @ p["a/a/kernel"] + p["a/a/bias"]
    return z if hidden_only else torch.tanh(z)


def t_critic(p, s, a):
    h = relu6(s @ p["l1/w1_s"] + a @ p["l1/w1_a"] + p["l1/b1"])
    h = relu6(h @ p["l2/kernel"] + p["l2/bias"])
    h = torch.relu(h @ p["l3/kernel"] + p["l3/bias"])
    return h @ p["q/dense/kernel"] + p["q/dense/bias"]


def t_gen(p, s, a):
    h = torch.relu(s @ p["l1/w1_s"] + a @ p["l1/w1_a"] + p["l1/b1"])
    h = torch.relu(h @ p["l2/kernel"] + p["l2/bias"])
    h = torch.relu(h @ p["l3/kernel"] + p["l3/bias"])
    return h @ p["l4/kernel"] + p["l4/bias"]


def t_disc(p, s, x, a):
    h = lrelu(s @ p["l1/w1_s"] + x @ p["l1/w1_G"] + a @ p["l1/w1_a"] + p["l1/b1"])
    h = lrelu(h @ p["l2/kernel"] + p["l2/bias"])
    h = lrelu(h @ p["l3/kernel"] + p["l3/bias"])
    return h @ p["D/w3"] + p["D/b3"]


class FedTanh(torch.autograd.Function):
    """tanh whose OUTPUT tensor is replaced by a fed value: forward returns the fed value,
    backward is TanhGrad on that value (what feeding `actor.a` does in TF1)."""

    @staticmethod
    def forward(ctx, z, fed):
        ctx.save_for_backward(fed)
        return fed.clone()

    @staticmethod
    def backward(ctx, g):
        (fed,) = ctx.saved_tensors
        return g * (1 - fed * fed), None


def flat_grads(spec, p):
    return np.concatenate([p[n].grad.numpy().reshape(-1) if p[n].grad is not None
                           else np.zeros(int(np.prod(s))) for n, s in spec.tensors])


@pytest.fixture
def agent_and_batch():
    ag = CuriousAgent(seed=1234, dtype=np.float64)
    rng = np.random.RandomState(0)
    B = 24
    s = rng.uniform(-1.5, 1.5, (B, 7)); s_ = rng.uniform(-1.5, 1.5, (B, 7)); g = rng.uniform(-1.5, 1.5, (B, 7))
    a = rng.uniform(-1, 1, (B, 2)); r = rng.uniform(0, 1, (B, 1))
    # scale some weights up so relu6's upper clip and lrelu's negative side are exercised
    ag.actor_e["l1/kernel"] *= 8; ag.critic_e["l1/w1_s"] *= 8; ag.actor_t["l1/kernel"] *= 8
    return ag, (s, a, r, s_, g)


def test_forward_matches_torch(agent_and_batch):
    ag, (s, a, r, s_, g) = agent_and_batch
    ts, ta, tg = map(torch.tensor, (s, a, g))
    np.testing.assert_allclose(ag.choose_action(s), t_actor(tparams(ag.actor_e), ts).detach(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ag.predict(s, a), t_gen(tparams(ag.gen), ts, ta).detach(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ag.determine(s, a, g), 1 - torch.sigmoid(t_disc(tparams(ag.disc), ts, tg, ta)).detach(),
                               rtol=1e-12, atol=1e-14)
    z1 = s @ ag.actor_e["l1/kernel"] + ag.actor_e["l1/bias"]
    assert (z1 > 6).any() and (z1 < 0).any()          # the clip regions are really hit


def test_critic_step_gradients(agent_and_batch):
    ag, (s, a, r, s_, g) = agent_and_batch
    out = ag.critic_grads(s, a, r, s_)
    ts, ta, tr, ts_ = map(torch.tensor, (s, a, r, s_))
    pc, pa = tparams(ag.critic_e), tparams(ag.actor_e)
    with torch.no_grad():
        y = tr + 0.9 * t_critic(tparams(ag.critic_t), ts_, t_actor(tparams(ag.actor_t), ts_))
    a_fed = FedTanh.apply(t_actor(pa, ts, hidden_only=True), ta)
    q = t_critic(pc, ts, a_fed)
    loss = torch.mean((y - q) ** 2)
    loss.backward()
    assert out["loss"] == pytest.approx(loss.item(), rel=1e-12)
    np.testing.assert_allclose(out["g_critic"], flat_grads(ag.critic_spec, pc), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["g_actor"], flat_grads(ag.actor_spec, pa), rtol=1e-9, atol=1e-14)
    assert np.abs(out["g_actor"]).max() > 0


def test_actor_step_gradients(agent_and_batch):
    ag, (s, a, r, s_, g) = agent_and_batch
    out = ag.actor_grads(s)
    ts = torch.tensor(s)
    pc, pa = tparams(ag.critic_e), tparams(ag.actor_e)
    act = t_actor(pa, ts)
    q = t_critic(pc, ts, act)
    (a_grads,) = torch.autograd.grad(q.sum(), act, retain_graph=True)      # tf.gradients(q, a)
    act.backward(a_grads)                                                  # tf.gradients(a, params, grad_ys)
    np.testing.assert_allclose(out["a_grads"], a_grads.numpy(), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["g_actor"], flat_grads(ag.actor_spec, pa), rtol=1e-9, atol=1e-14)


def test_disc_and_gen_step_gradients(agent_and_batch):
    ag, (s, a, r, s_, g) = agent_and_batch
    ts, ta, ts_, tg = map(torch.tensor, (s, a, s_, g))
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    out = ag.disc_grads(g, s_, s, a)
    pd = tparams(ag.disc)
    lf, lr = t_disc(pd, ts, tg, ta), t_disc(pd, ts, ts_, ta)
    loss = bce(lr, torch.ones_like(lr)) + bce(lf, torch.zeros_like(lf))
    loss.backward()
    assert out["loss"] == pytest.approx(loss.item(), rel=1e-12)
    assert ag.disc_eval(g, s_, s, a) == pytest.approx(loss.item(), rel=1e-12)
    np.testing.assert_allclose(out["g_disc"], flat_grads(ag.disc_spec, pd), rtol=1e-9, atol=1e-14)

    out = ag.gen_grads(s, a)
    pd, pg = tparams(ag.disc), tparams(ag.gen)
    lf = t_disc(pd, ts, t_gen(pg, ts, ta), ta)
    loss = bce(lf, torch.ones_like(lf))
    loss.backward()
    assert out["loss"] == pytest.approx(loss.item(), rel=1e-12)
    assert ag.gen_eval(s, a) == pytest.approx(loss.item(), rel=1e-12)
    np.testing.assert_allclose(out["g_gen"], flat_grads(ag.gen_spec, pg), rtol=1e-9, atol=1e-14)


def test_lrelu_tie_goes_to_alpha_branch():
    z = np.array([[-1.0, 0.0, 2.0]])
    np.testing.assert_array_equal(nets.act_bwd("lrelu", z, nets.act_fwd("lrelu", z), np.ones_like(z)), [[0.2, 0.2, 1.0]])
    z6 = np.array([[0.0, 3.0, 6.0, 7.0]])
    np.testing.assert_array_equal(nets.act_bwd("relu6", z6, nets.act_fwd("relu6", z6), np.ones_like(z6)), [[0, 1, 0, 0]])


def test_optimizers_against_torch_with_matched_eps_form():
    rng = np.random.RandomState(3)
    w0 = rng.randn(50)
    grads = [rng.randn(50) for _ in range(5)]
    # RMSProp: TF1 puts eps inside the sqrt and starts ms at one; torch puts eps outside.
    tw = torch.tensor(w0.copy(), requires_grad=True)
    topt = torch.optim.RMSprop([tw], lr=1e-2, alpha=0.9, eps=0.0)
    opt = RMSProp(50, 1e-2, eps=0.0)
    w = w0.copy()
    for i, g in enumerate(grads):
        tw.grad = torch.tensor(g)
        if i == 0:
            topt.step()            # creates state; redo this step with square_avg starting at one
            with torch.no_grad():
                tw.copy_(torch.tensor(w0))
            st = topt.state[tw]
            st["square_avg"].fill_(1.0)
        topt.step()
        w = opt.step(w, g)
        np.testing.assert_allclose(w, tw.detach().numpy(), rtol=1e-12)
    # Adam: TF1's lr_t form == torch's bias-corrected form up to where eps enters
    tw = torch.tensor(w0.copy(), requires_grad=True)
    topt = torch.optim.Adam([tw], lr=1e-3, betas=(0.9, 0.999), eps=0.0)
    opt = Adam(50, 1e-3, eps=0.0)
    w = w0.copy()
    for g in grads:
        tw.grad = torch.tensor(g)
        topt.step()
        w = opt.step(w, g)
        np.testing.assert_allclose(w, tw.detach().numpy(), rtol=1e-10)


def test_tf1_slot_conventions():
    o = RMSProp(3, 1e-4)
    assert (o.ms == 1).all() and (o.mom == 0).all() and o.eps == 1e-10 and o.decay == 0.9
    w = o.step(np.zeros(3), np.full(3, 2.0))
    np.testing.assert_allclose(o.ms, 0.9 + 0.1 * 4)
    np.testing.assert_allclose(w, -1e-4 * 2 / np.sqrt(1.3 + 1e-10))
    a = Adam(3, 1e-4)
    w = a.step(np.zeros(3), np.full(3, 2.0))
    lr1 = 1e-4 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(w, -lr1 * 0.2 / (np.sqrt(0.004) + 1e-8))


def test_target_sync_schedule_and_independent_init():
    ag = CuriousAgent(seed=7, hyper=Hyper(replace_iter_a=3, replace_iter_c=2))
    assert not np.array_equal(ag.flat("actor_e"), ag.flat("actor_t"))      # separate xavier draws
    rng = np.random.RandomState(0)
    B = 8
    batch = rng.uniform(-1, 1, (B, 25))
    sync_c, sync_a = [], []
    for it in range(6):
        ag.update(batch)
        sync_c.append(np.array_equal(ag.flat("critic_e"), ag.flat("critic_t")))
        sync_a.append(np.array_equal(ag.flat("actor_e"), ag.flat("actor_t")))
    assert sync_c == [True, False, True, False, True, False]
    # the actor eval net is also moved by the critic step that follows the copy
    assert sync_a[0] and sync_a[3]


def test_data_parallel_contract_sum_of_shards_equals_full_batch():
    ag = CuriousAgent(seed=5)
    rng = np.random.RandomState(1)
    B, N = 32, 4
    s = rng.uniform(-1, 1, (B, 7)); s_ = rng.uniform(-1, 1, (B, 7)); g = rng.uniform(-1, 1, (B, 7))
    a = rng.uniform(-1, 1, (B, 2)); r = rng.uniform(0, 1, (B, 1))
    sh = [slice(i * B // N, (i + 1) * B // N) for i in range(N)]
    full = ag.critic_grads(s, a, r, s_)
    parts = [ag.critic_grads(s[k], a[k], r[k], s_[k], global_batch=B) for k in sh]
    np.testing.assert_allclose(sum(p["g_critic"] for p in parts), full["g_critic"], rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(sum(p["g_actor"] for p in parts), full["g_actor"], rtol=1e-10, atol=1e-15)
    full = ag.actor_grads(s)
    parts = [ag.actor_grads(s[k]) for k in sh]                              # batch SUM: no scaling
    np.testing.assert_allclose(sum(p["g_actor"] for p in parts), full["g_actor"], rtol=1e-10, atol=1e-15)
    full = ag.disc_grads(g, s_, s, a)
    parts = [ag.disc_grads(g[k], s_[k], s[k], a[k], global_batch=B) for k in sh]
    np.testing.assert_allclose(sum(p["g_disc"] for p in parts), full["g_disc"], rtol=1e-10, atol=1e-15)
    full = ag.gen_grads(s, a)
    parts = [ag.gen_grads(s[k], a[k], global_batch=B) for k in sh]
    np.testing.assert_allclose(sum(p["g_gen"] for p in parts), full["g_gen"], rtol=1e-10, atol=1e-15)
