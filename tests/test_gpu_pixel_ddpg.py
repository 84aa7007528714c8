"""Actor and critic of the pixel variant on the device (Conv_model.py:569-784) against the float64 restatement:
actions, q values, the TD loss, every gradient entry (1e-5 relative to the tensor scale, fp32), the RMSProp updates
given those gradients, and the target copies on the reference's schedule."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H = 256


def _f32(v):
    return np.asarray(v).astype(np.float32).astype(np.float64)


def _params(rng, ad):
    from test_oracle_pixel_ddpg import params
    A, C = params(rng, H, ad, spread=0.5)
    return {k: _f32(v) for k, v in A.items()}, {k: _f32(v) for k, v in C.items()}


class _Enc:
    """Stands in for the state encoder where the test feeds features directly."""
    features = H


def _setup(ad, seed=0, period_a=3, period_c=2, encoder=None):
    import rl_gan_amd as R
    rng = np.random.RandomState(seed)
    A, C = _params(rng, ad)
    At, Ct = _params(rng, ad)
    enc = encoder or _Enc()
    actor = R.PixelActor(ad, 1e-4, period_a, enc, params=A, target_params=At)
    critic = R.PixelCritic(ad, 1e-4, 0.99, period_c, actor, enc, params=C, target_params=Ct)
    return actor, critic, A, C, At, Ct, rng


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def _layer_grads(net, names, shapes):
    """Per layer (kernel, bias) gradients cut back to the reference's shapes; asserts that the padding got none."""
    out = {}
    for L, (kn, bn), (k, n) in zip(net.layers, names, shapes):
        g = L.grad.cpu().numpy()
        W, b = g[:L.k * L.n].reshape(L.k, L.n), g[L.k * L.n:]
        assert not W[k:].any() and not W[:, n:].any() and not b[n:].any(), kn
        out[kn], out[bn] = W[:k, :n], b[:n]
    return out


def _flat(P, keys_shapes, layer):
    """The layer's canonical parameter vector (padded) from reference-shaped arrays."""
    k = np.zeros((layer.k, layer.n)); b = np.zeros(layer.n)
    (kern, bias) = keys_shapes
    k[:kern.shape[0], :kern.shape[1]] = kern; b[:bias.size] = bias.reshape(-1)
    return np.concatenate([k.ravel(), b]).astype(np.float32)


@pytest.mark.parametrize("n,ad", [(16, 6), (37, 14), (300, 3)])
def test_critic_step(n, ad):
    from gpu_util import close
    from oracle import pixel_ddpg as od
    from oracle.optim import RMSProp
    actor, critic, A, C, At, Ct, rng = _setup(ad)
    phis, phis_n = _f32(rng.uniform(0, 2, (n, H))), _f32(rng.uniform(0, 2, (n, H)))
    a, r = np.eye(ad)[rng.randint(0, ad, n)], _f32(rng.uniform(0, 1, (n, 1)))
    close(actor.actions(_dev(phis_n), target=True).cpu().numpy()[:, :ad], od.actor_forward(phis_n, At)[0], what="a'")
    close(critic.q_batch(_dev(phis), _dev(a)).cpu().numpy(), od.critic_forward(phis, a, C)[0], what="q")
    loss, G = od.critic_loss_and_grads(phis, a, r, phis_n, C, Ct, At, 0.99)
    ev = critic.eval_net.layers
    before = [L.canon.cpu().numpy().copy() for L in ev]
    t_before = [L.canon.cpu().numpy().copy() for L in critic.target_net.layers]
    got = float(critic.learn(None, _dev(a), _dev(r), None, phis=_dev(phis), phis_next=_dev(phis_n)).item())
    assert abs(got - loss) <= 1e-5 * loss
    g1 = ev[0].grad.cpu().numpy()
    W = g1[:critic.k_pad * 200].reshape(critic.k_pad, 200)
    close(W[:H], G["w1_s"], what="dw1_s"); close(W[H:H + ad], G["w1_a"], what="dw1_a"); assert not W[H + ad:].any()
    close(g1[critic.k_pad * 200:], G["b1"], what="db1")
    got_g = _layer_grads(critic.eval_net, [("w1", "b1"), ("w2", "b2"), ("w3", "b3"), ("wq", "bq")],
                         [(critic.k_pad, 200), (200, 200), (200, 10), (10, 1)])
    for k in ("w2", "b2", "w3", "b3", "wq", "bq"):
        close(got_g[k], G[k].reshape(got_g[k].shape), what="d" + k)
    for L, w0 in zip(ev, before):
        want = RMSProp(w0.size, 1e-4, dtype=np.float32).step(w0, L.grad.cpu().numpy())
        np.testing.assert_allclose(L.canon.cpu().numpy(), want, rtol=0, atol=2e-7)
    # counter 0: the copy follows the first step (Conv_model.py:661-663); the second call (period 2) leaves the target alone
    for Lt, Le in zip(critic.target_net.layers, ev):
        assert torch.equal(Lt.canon, Le.canon)
    snap = [L.canon.clone() for L in critic.target_net.layers]
    critic.learn(None, _dev(a), _dev(r), None, phis=_dev(phis), phis_next=_dev(phis_n))
    for Lt, Le, s0, tb in zip(critic.target_net.layers, ev, snap, t_before):
        assert torch.equal(Lt.canon, s0) and not torch.equal(Lt.canon, Le.canon) and not np.array_equal(tb, s0.cpu().numpy())
    critic.learn(None, _dev(a), _dev(r), None, phis=_dev(phis), phis_next=_dev(phis_n))
    for Lt, Le in zip(critic.target_net.layers, ev):
        assert torch.equal(Lt.canon, Le.canon)
    assert int(critic.t_replace_counter.item()) == 3


@pytest.mark.parametrize("n,ad", [(16, 6), (37, 14), (300, 3)])
def test_actor_step(n, ad):
    from gpu_util import close
    from oracle import pixel_ddpg as od
    from oracle.optim import RMSProp
    actor, critic, A, C, At, Ct, rng = _setup(ad, seed=1)
    phis = _f32(rng.uniform(0, 2, (n, H)))
    a, da, G = od.actor_policy_grads(phis, A, C)
    ev = actor.eval_net.layers
    before = [L.canon.cpu().numpy().copy() for L in ev]
    a_dev = actor.actions(_dev(phis)).clone()                        # the action the step is about to differentiate
    actor.learn(None, critic, phis=_dev(phis))
    assert actor.row_passes == (ad <= 12)                            # (37, 14): 16 action columns, the layer-by-layer form
    close(a_dev.cpu().numpy()[:, :ad], a, what="a")
    assert not a_dev[:, ad:].any()
    got_g = _layer_grads(actor.eval_net, [("w1", "b1"), ("w2", "b2"), ("w3", "b3"), ("wa", "ba")], [(H, 200), (200, 200), (200, 10), (10, ad)])
    for k in G:
        close(got_g[k], G[k].reshape(got_g[k].shape), what="d" + k)
    for L, w0 in zip(ev, before):
        want = RMSProp(w0.size, -1e-4, dtype=np.float32).step(w0, L.grad.cpu().numpy())       # ascent: RMSProp(-lr)
        np.testing.assert_allclose(L.canon.cpu().numpy(), want, rtol=0, atol=2e-7)
    for Lt, Le in zip(actor.target_net.layers, ev):
        assert torch.equal(Lt.canon, Le.canon)
    # the target scope's LSTM cell is in t_params too (Conv_model.py:708-709,752-753): call 0 copied the eval cell over it
    assert torch.equal(actor.LSTM_unit_.cell.canon, actor.LSTM_unit.cell.canon)
    actor.LSTM_unit.cell.canon.mul_(0.5)                             # stands for a restored checkpoint changing the eval cell
    for _ in range(2):
        actor.learn(None, critic, phis=_dev(phis))
    for Lt, Le in zip(actor.target_net.layers, ev):
        assert not torch.equal(Lt.canon, Le.canon)                   # period 3: calls 1 and 2 do not copy
    assert not torch.equal(actor.LSTM_unit_.cell.canon, actor.LSTM_unit.cell.canon)
    actor.learn(None, critic, phis=_dev(phis))
    for Lt, Le in zip(actor.target_net.layers, ev):
        assert torch.equal(Lt.canon, Le.canon)
    assert torch.equal(actor.LSTM_unit_.cell.canon, actor.LSTM_unit.cell.canon)


@pytest.mark.parametrize("n", [16, 100, 4096])
def test_row_passes_against_the_layer_by_layer_form(n):
    """The two forms of the steps -- one row-pass launch over 16-row tiles, or a matrix product per layer -- from the
    same parameters on the same minibatch (full tiles, a ragged last tile, the bench's batch): loss, every gradient, the
    updated parameters, optimizer slots and target copies agree to float32 rounding; three steps in a row."""
    from gpu_util import close
    from oracle import pixel_ddpg as od
    ad = 6
    rng = np.random.RandomState(9)
    phis64, phis_n64 = _f32(rng.uniform(0, 2, (n, H))), _f32(rng.uniform(0, 2, (n, H)))
    a64, r64 = np.eye(ad)[rng.randint(0, ad, n)], _f32(rng.uniform(0, 1, (n, 1)))
    phis, phis_n, a, r = _dev(phis64), _dev(phis_n64), _dev(a64), _dev(r64)
    pairs = []
    for rows in (True, False):
        actor, critic, A, Cp, At, Ct, _ = _setup(ad, seed=5)
        assert actor.row_passes and critic.row_passes
        actor.row_passes = critic.row_passes = rows
        pairs.append((actor, critic))
    loss64, G64 = od.critic_loss_and_grads(phis64, a64, r64, phis_n64, Cp, Ct, At, 0.99)
    for it in range(3):
        losses = [float(c.learn(None, a, r, None, phis=phis, phis_next=phis_n).item()) for _, c in pairs]
        assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[1])
        if it == 0:                                                  # the row passes against the float64 restatement, the usual bar
            assert abs(losses[0] - loss64) <= 1e-5 * loss64
            c = pairs[0][1]
            g1 = c.eval_net.layers[0].grad.cpu().numpy()
            W = g1[:c.k_pad * 200].reshape(c.k_pad, 200)
            close(W[:H], G64["w1_s"], what="dw1_s"); close(W[H:H + ad], G64["w1_a"], what="dw1_a"); close(g1[c.k_pad * 200:], G64["b1"], what="db1")
            got = _layer_grads(c.eval_net, [("w1", "b1"), ("w2", "b2"), ("w3", "b3"), ("wq", "bq")], [(c.k_pad, 200), (200, 200), (200, 10), (10, 1)])
            for k in ("w2", "b2", "w3", "b3", "wq", "bq"):
                close(got[k], G64[k].reshape(got[k].shape), what="critic d" + k)
        for act, c in pairs:
            act.learn(None, c, phis=phis)
        for (a0, c0), (a1, c1) in [pairs]:
            for net0, net1, what in ((c0, c1, "critic"), (a0, a1, "actor")):
                for i, (L0, L1) in enumerate(zip(net0.eval_net.layers, net1.eval_net.layers)):
                    # Two float32 evaluations: a few 1e-7 apart in every sum, and at 4096 rows some hidden unit of some row
                    # has its pre-activation within rounding of 0 or 6, where relu6' differs between the two (measured
                    # here: 8.6e-5 of the tensor's scale in dW2 of the layer-by-layer form against the float64
                    # restatement, the row passes within 3e-7).  One such row is 1/4096 of a column's terms.
                    close(L0.grad.cpu().numpy(), L1.grad.cpu().numpy().astype(np.float64), rtol=5e-5 if n <= 100 else 1e-3,
                          what=f"{what} layer {i} gradient, step {it}")
                    # an RMSProp step moves a weight by about lr = 1e-4 whatever its gradient's size: the two forms' steps
                    # differ by the gradients' relative difference
                    wtol = (5e-7 if n <= 100 else 2e-6) * (it + 1)
                    np.testing.assert_allclose(L0.canon.cpu().numpy(), L1.canon.cpu().numpy(), rtol=0, atol=wtol)
                for o0, o1 in zip(net0._opt, net1._opt):
                    np.testing.assert_allclose(o0.ms.cpu().numpy(), o1.ms.cpu().numpy(), rtol=1e-5 if n <= 100 else 2e-3, atol=1e-9)
                for T0, T1 in zip(net0.target_net.layers, net1.target_net.layers):
                    np.testing.assert_allclose(T0.canon.cpu().numpy(), T1.canon.cpu().numpy(), rtol=0, atol=wtol)
            assert int(c0.t_replace_counter.item()) == int(c1.t_replace_counter.item()) == it + 1
    # the layer-by-layer calls read the same parameters the row passes wrote: the layers are views of the flat vector
    a0, c0 = pairs[0]
    np.testing.assert_allclose(a0.actions(phis).cpu().numpy(), pairs[1][0].actions(phis).cpu().numpy(), rtol=0, atol=2e-5)


def test_steps_from_frames_and_action_choice():
    """Both steps with the state encoder in front (frames in, features computed on the device) and the single-state
    action choice with the reference's NumPy noise."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import pixel_ddpg as od
    from oracle import pixel_gan as op
    rng = np.random.RandomState(4)
    u = lambda lim, shape: _f32(rng.uniform(-lim, lim, shape))
    n, ad, feat = 5, 6, 9 * 9 * 32
    st_params = [(u(0.2, (3, 3, 4, 32)), u(0.1, 32)), (u(0.1, (3, 3, 32, 32)), u(0.1, 32))]
    st_dense = (u(0.03, (feat, 256)), _f32(np.full(256, 0.001)))
    enc = R.ConvEncoder(42, 42, 4, R.universe_head(2), params=st_params, dense=256, dense_params=st_dense)
    enc_before = [L.canon.clone() for L in enc.layers + [enc.head]]
    actor, critic, A, C, At, Ct, _ = _setup(ad, seed=2, encoder=enc)
    s, s_n = _f32(rng.randint(0, 256, (n, 42, 42, 4)) / 255.0), _f32(rng.randint(0, 256, (n, 42, 42, 4)) / 255.0)
    a, r = np.eye(ad)[rng.randint(0, ad, n)], _f32(rng.uniform(0, 1, (n, 1)))
    lay = op.oc.universe_head(2)
    phis, _ = op.encoder(s, st_params, st_dense, lay)
    phis_n, _ = op.encoder(s_n, st_params, st_dense, lay)
    loss, Gc = od.critic_loss_and_grads(phis, a, r, phis_n, C, Ct, At, 0.99)
    # the action choice before any step: same NumPy draw on both sides
    np.random.seed(11)
    got = actor.choose_action(s[0], None, 0.3)
    np.random.seed(11)
    noisy = np.random.normal(od.actor_forward(phis[:1], A)[0], 0.3)
    want = np.zeros(ad, dtype=np.int64); want[noisy.argmax(axis=1)] = 1
    assert np.array_equal(got, want) and got.sum() == 1
    # the state path (Curious_GAN_mario.py:69): two steps from the initial state on the first frame stack
    unit = actor.LSTM_unit
    kern, bias = unit.cell.kernel_bias()
    st_o, st_d = [np.zeros((1, H)), np.zeros((1, H))], unit.get_initial_state()
    for _ in range(2):
        st_o = list(od.lstm_step(phis[:1], st_o[0], st_o[1], kern.astype(np.float64), bias.astype(np.float64)))
        st_d = unit.get_state(s[0], st_d)
    close(st_d[0], st_o[0], what="lstm c"); close(st_d[1], st_o[1], what="lstm h")
    got_loss = float(critic.learn(_dev(s), _dev(a), _dev(r), _dev(s_n)).item())
    assert abs(got_loss - loss) <= 1e-5 * loss
    g2 = critic.eval_net.layers[1].grad.cpu().numpy()
    close(g2[:200 * 200].reshape(200, 200), Gc["w2"], what="critic dw2 from frames")
    # the actor's step sees the critic as updated by the step above
    Cn = {}
    L1 = critic.eval_net.layers[0].kernel_bias()
    Cn["w1_s"], Cn["w1_a"], Cn["b1"] = L1[0][:H].astype(np.float64), L1[0][H:H + ad].astype(np.float64), L1[1].astype(np.float64)
    k, b = critic.eval_net.layers[1].kernel_bias(); Cn["w2"], Cn["b2"] = k.astype(np.float64), b.astype(np.float64)
    k, b = critic.eval_net.layers[2].kernel_bias(); Cn["w3"], Cn["b3"] = k[:, :10].astype(np.float64), b[:10].astype(np.float64)
    k, b = critic.eval_net.layers[3].kernel_bias(); Cn["wq"], Cn["bq"] = k[:10, :1].astype(np.float64), b[:1].astype(np.float64)
    _, _, Ga = od.actor_policy_grads(phis, A, Cn)
    actor.learn(_dev(s), critic)
    g1 = actor.eval_net.layers[0].grad.cpu().numpy()
    close(g1[:H * 200].reshape(H, 200), Ga["w1"], what="actor dw1 from frames")
    for L, w0 in zip(enc.layers + [enc.head], enc_before):       # neither step trains the encoder
        assert torch.equal(L.canon, w0)


@pytest.mark.parametrize("n", [1, 33])
def test_lstm_state_step(n):
    """The state path of the pixel actor (`lstm_unit.get_state`, Conv_model.py:67-99) against the restatement."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import pixel_ddpg as od
    rng = np.random.RandomState(n)
    kernel, bias = _f32(rng.uniform(-0.08, 0.08, (2 * H, 4 * H))), _f32(rng.uniform(-0.2, 0.2, 4 * H))
    unit = R.LSTMUnit(_Enc(), kernel=kernel, bias=bias)
    phis, c, h = _f32(rng.uniform(0, 2, (n, H))), _f32(rng.standard_normal((n, H))), _f32(rng.uniform(-1, 1, (n, H)))
    c2, h2 = od.lstm_step(phis, c, h, kernel, bias)
    gc, gh = unit.step_batch(_dev(phis), _dev(c), _dev(h))
    close(gc.cpu().numpy(), c2, what="c'"); close(gh.cpu().numpy(), h2, what="h'")
    init = unit.get_initial_state()
    assert len(init) == 2 and init[0].shape == (1, H) and not init[0].any() and not init[1].any()
