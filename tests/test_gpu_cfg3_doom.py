"""The pixel DDPG configuration as composed: the four-convolution "doom" head on 84x84x4 frames (8@5x5/4, 16@3x3/2,
32@3x3/2, 64@3x3/2, ELU, Conv_DDPG.py:83-86 / Conv_model.py:15-22) -> 64 features -> the 256-unit ELU layer
(Conv_model.py:24) -> the actor's and the critic's update steps (Conv_model.py:569-784).  A small batch against the
float64 restatement (loss, every gradient tensor, the RMSProp updates), and the full batch of 4096 frames through
properties that do not need the restatement: finite, bit-for-bit repeatable, target copies on their periods."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H = 256


def _f32(v):
    return np.asarray(v).astype(np.float32).astype(np.float64)


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def _encoder_params(rng):
    u = lambda lim, shape: _f32(rng.uniform(-lim, lim, shape))
    convs = [(u(0.25, (5, 5, 4, 8)), u(0.1, 8)), (u(0.3, (3, 3, 8, 16)), u(0.1, 16)), (u(0.25, (3, 3, 16, 32)), u(0.1, 32)),
             (u(0.2, (3, 3, 32, 64)), u(0.1, 64))]
    dense = (u(0.15, (64, H)), _f32(np.full(H, 0.001)))
    return convs, dense


def _critic_dict(critic, ad):
    C = {}
    k, b = critic.eval_net.layers[0].kernel_bias()
    C["w1_s"], C["w1_a"], C["b1"] = k[:H].astype(np.float64), k[H:H + ad].astype(np.float64), b.astype(np.float64)
    k, b = critic.eval_net.layers[1].kernel_bias(); C["w2"], C["b2"] = k.astype(np.float64), b.astype(np.float64)
    k, b = critic.eval_net.layers[2].kernel_bias(); C["w3"], C["b3"] = k[:, :10].astype(np.float64), b[:10].astype(np.float64)
    k, b = critic.eval_net.layers[3].kernel_bias(); C["wq"], C["bq"] = k[:10, :1].astype(np.float64), b[:1].astype(np.float64)
    return C


@pytest.mark.parametrize("n,ad", [(6, 6), (19, 3)])
def test_doom_head_actor_and_critic_steps_from_frames(n, ad):
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import pixel_ddpg as od
    from oracle import pixel_gan as op
    from oracle.optim import RMSProp
    from test_gpu_pixel_ddpg import _layer_grads, _setup
    rng = np.random.RandomState(30 + n)
    convs, dense = _encoder_params(rng)
    enc = R.ConvEncoder(84, 84, 4, R.DOOM_HEAD, params=convs, dense=H, dense_params=dense)
    assert enc.features == H and [L.geom.h_out for L in enc.layers] == [20, 9, 4, 1]
    enc_before = [L.canon.clone() for L in enc.layers + [enc.head]]
    actor, critic, A, C, At, Ct, _ = _setup(ad, seed=7, encoder=enc)
    s, s_n = _f32(rng.uniform(0, 1, (n, 84, 84, 4))), _f32(rng.uniform(0, 1, (n, 84, 84, 4)))
    a, r = np.eye(ad)[rng.randint(0, ad, n)], _f32(rng.uniform(0, 1, (n, 1)))
    phis, _ = op.encoder(s, convs, dense, op.oc.DOOM_HEAD)
    phis_n, _ = op.encoder(s_n, convs, dense, op.oc.DOOM_HEAD)
    close(enc.forward(_dev(s)).cpu().numpy(), phis, what="doom-head features")
    # critic.learn from frames
    loss, Gc = od.critic_loss_and_grads(phis, a, r, phis_n, C, Ct, At, 0.99)
    before = [L.canon.cpu().numpy().copy() for L in critic.eval_net.layers]
    got_loss = float(critic.learn(_dev(s), _dev(a), _dev(r), _dev(s_n)).item())
    assert abs(got_loss - loss) <= 1e-5 * loss
    g1 = critic.eval_net.layers[0].grad.cpu().numpy()
    W = g1[:critic.k_pad * 200].reshape(critic.k_pad, 200)
    close(W[:H], Gc["w1_s"], what="dw1_s"); close(W[H:H + ad], Gc["w1_a"], what="dw1_a"); close(g1[critic.k_pad * 200:], Gc["b1"], what="db1")
    got = _layer_grads(critic.eval_net, [("w1", "b1"), ("w2", "b2"), ("w3", "b3"), ("wq", "bq")],
                       [(critic.k_pad, 200), (200, 200), (200, 10), (10, 1)])
    for k in ("w2", "b2", "w3", "b3", "wq", "bq"):
        close(got[k], Gc[k].reshape(got[k].shape), what="critic d" + k)
    for L, w0 in zip(critic.eval_net.layers, before):
        want = RMSProp(w0.size, 1e-4, dtype=np.float32).step(w0, L.grad.cpu().numpy())
        np.testing.assert_allclose(L.canon.cpu().numpy(), want, rtol=0, atol=2e-7)
    # actor.learn from frames, through the critic as the step above left it
    _, _, Ga = od.actor_policy_grads(phis, A, _critic_dict(critic, ad))
    before = [L.canon.cpu().numpy().copy() for L in actor.eval_net.layers]
    actor.learn(_dev(s), critic)
    got = _layer_grads(actor.eval_net, [("w1", "b1"), ("w2", "b2"), ("w3", "b3"), ("wa", "ba")], [(H, 200), (200, 200), (200, 10), (10, ad)])
    for k in Ga:
        close(got[k], Ga[k].reshape(got[k].shape), what="actor d" + k)
    for L, w0 in zip(actor.eval_net.layers, before):
        want = RMSProp(w0.size, -1e-4, dtype=np.float32).step(w0, L.grad.cpu().numpy())
        np.testing.assert_allclose(L.canon.cpu().numpy(), want, rtol=0, atol=2e-7)
    for L, w0 in zip(enc.layers + [enc.head], enc_before):           # neither step trains the encoder (Conv_model.py:616,784)
        assert torch.equal(L.canon, w0)


def test_doom_head_ddpg_update_at_batch_4096():
    """Synthetic frames, uniform [0, 1) float32 from seed 0, batch 4096: two runs from the same weights give the same bits,
    everything stays finite, and the targets are copied on calls 0 and 2 (critic, period 2) / 0 and 3 (actor, period 3)."""
    import rl_gan_amd as R
    from test_gpu_pixel_ddpg import _setup
    B, ad = 4096, 6
    g = torch.Generator(device="cuda").manual_seed(0)
    s = torch.rand((B, 84, 84, 4), generator=g, device="cuda", dtype=torch.float32)
    s_n = torch.rand((B, 84, 84, 4), generator=g, device="cuda", dtype=torch.float32)
    a = torch.nn.functional.one_hot(torch.randint(0, ad, (B,), generator=g, device="cuda"), ad).float()
    r = torch.rand((B, 1), generator=g, device="cuda", dtype=torch.float32)
    runs = []
    for _ in range(2):
        convs, dense = _encoder_params(np.random.RandomState(1))
        enc = R.ConvEncoder(84, 84, 4, R.DOOM_HEAD, params=convs, dense=H, dense_params=dense)
        actor, critic, *_ = _setup(ad, seed=3, encoder=enc)
        losses, copied_c, copied_a = [], [], []
        for it in range(4):
            losses.append(float(critic.learn(s, a, r, s_n).item()))
            actor.learn(s, critic)
            copied_c.append(all(torch.equal(t.canon, e.canon) for t, e in zip(critic.target_net.layers, critic.eval_net.layers)))
            copied_a.append(all(torch.equal(t.canon, e.canon) for t, e in zip(actor.target_net.layers, actor.eval_net.layers)))
        assert copied_c == [True, False, True, False] and copied_a == [True, False, False, True]
        assert int(critic.t_replace_counter.item()) == 4 and int(actor.t_replace_counter.item()) == 4
        w = torch.cat([L.canon for L in critic.eval_net.layers + actor.eval_net.layers])
        assert np.isfinite(losses).all() and bool(torch.isfinite(w).all())
        assert losses[0] > 0 and len(set(losses)) == 4               # the critic moves: four different losses
        runs.append((losses, w.cpu().numpy()))
    assert runs[0][0] == runs[1][0]
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
