"""The pixel discriminator's training step on the device (Conv_model.py:159-340, Curious_GAN_mario.py:340-346)
against the float64 restatement: both losses, the intrinsic reward, every gradient of the discriminator's and
the two encoders' variables (1e-5 relative to the tensor scale, fp32), and the Adam update given those gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n, a_dim, seed=0):
    import rl_gan_amd as R
    rng = np.random.RandomState(seed)
    f32 = lambda v: np.asarray(v).astype(np.float32).astype(np.float64)
    u = lambda lim, shape: f32(rng.uniform(-lim, lim, shape))
    feat = 9 * 9 * 32                                       # 42 -> 20 -> 9
    st_params = [(u(0.2, (3, 3, 4, 32)), u(0.1, 32)), (u(0.1, (3, 3, 32, 32)), u(0.1, 32))]
    st_dense = (u(0.03, (feat, 256)), f32(np.full(256, 0.001)))
    enc_params = [(u(0.3, (3, 3, 1, 32)), u(0.1, 32)), (u(0.1, (3, 3, 32, 32)), u(0.1, 32))]
    enc_dense = (u(0.03, (feat, 256)), f32(np.full(256, 0.001)))
    P = dict(w1_G=u(0.1, (256, 200)), w1_s=u(0.1, (256, 200)), w1_a=u(0.2, (a_dim, 200)), b1=f32(np.full(200, 0.001)),
             w3=u(0.2, (200, 10)), b3=f32(np.full(10, 0.001)), w2=u(0.5, (10, 1)), b2=f32(np.full(1, 0.001)))
    s = f32(rng.randint(0, 256, (n, 42, 42, 4)) / 255.0)
    s_next = f32(rng.uniform(0, 1, (n, 42, 42, 1)))
    g_img = f32(rng.uniform(-0.2, 1.0, (n, 42, 42, 1)))
    a = f32(rng.uniform(-1, 1, (n, a_dim)))
    st_enc = R.ConvEncoder(42, 42, 4, R.universe_head(2), params=st_params, dense=256, dense_params=st_dense)
    ob_enc = R.single_frame_encoder(42, 2, params=enc_params, dense_params=enc_dense)
    D = R.PixelDiscriminator(a_dim, 1e-4, st_enc, ob_enc, params=P)
    host = dict(s=s, a=a, s_next=s_next, g_img=g_img, P=P, st_params=st_params, st_dense=st_dense, enc_params=enc_params,
                enc_dense=enc_dense)
    dev = {k: torch.from_numpy(host[k].astype(np.float32)).cuda() for k in ("s", "a", "s_next", "g_img")}
    return D, host, dev


@pytest.mark.parametrize("n,a_dim", [(5, 6), (3, 14)])
def test_discriminator_step_matches_restatement(n, a_dim):
    from gpu_util import close
    from oracle import pixel_gan as op
    from oracle.optim import Adam
    D, H, T = _setup(n, a_dim)
    lr_, lf_, g = op.discriminator_loss_and_grads(H["s"], H["a"], H["s_next"], H["g_img"], H["P"], H["st_params"], H["st_dense"],
                                                  H["enc_params"], H["enc_dense"])
    # intrinsic reward 1 - D_fake
    lay = op.oc.universe_head(2)
    phis, _ = op.encoder(H["s"], H["st_params"], H["st_dense"], lay)
    fobs, _ = op.encoder(H["g_img"], H["enc_params"], H["enc_dense"], lay)
    xf, _ = op.discriminator_net(phis, fobs, H["a"], H["P"])
    close(D.determine_batch(T["s"], T["a"], T["g_img"]).cpu().numpy(), 1.0 - 1.0 / (1.0 + np.exp(-xf)), what="1 - D_fake")
    assert abs(D.eval(T["g_img"], T["s_next"], T["s"], T["a"]) - (lr_ + lf_)) <= 1e-5 * (lr_ + lf_)
    trained = [o.layer for o in D._opt]
    before = [L.canon.cpu().numpy().copy() for L in trained]
    losses = D.learn(T["g_img"], T["s_next"], T["s"], T["a"]).cpu().numpy()
    np.testing.assert_allclose(losses, [lr_, lf_], rtol=1e-5)
    # discriminator layers (first layer rows: [w1_s, w1_G, w1_a, padding])
    g1 = D.l1.grad.cpu().numpy()
    k = D.k_pad
    W = g1[:k * 200].reshape(k, 200)
    close(W[:256], g["w1_s"], what="dw1_s"); close(W[256:512], g["w1_G"], what="dw1_G")
    close(W[512:512 + a_dim], g["w1_a"], what="dw1_a"); assert not W[512 + a_dim:].any()
    close(g1[k * 200:], g["b1"], what="db1")
    g3 = D.l3.grad.cpu().numpy()
    W3 = g3[:200 * 12].reshape(200, 12)
    close(W3[:, :10], g["w3"], what="dw3"); assert not W3[:, 10:].any()
    close(g3[200 * 12:][:10], g["b3"], what="db3")
    g2 = D.l2.grad.cpu().numpy()
    W2 = g2[:48].reshape(12, 4)
    close(W2[:10, :1], g["w2"], what="dw2"); assert not W2[:, 1:].any() and not W2[10:].any()
    close(g2[48:][:1], g["b2"], what="db2")
    # encoders
    for name, enc in (("state_encoder", D.state_encoder), ("obs_encoder", D.obs_encoder)):
        for i, (L, (dw, db)) in enumerate(zip(enc.layers + [enc.head], g[name])):
            gl = L.grad.cpu().numpy()
            got, gb = gl[:gl.size - db.size], gl[gl.size - db.size:]
            if name == "obs_encoder" and i == 0:            # kept as a 2x2 kernel over 2x2 blocks: seven entries stay zero
                got, gb = L.frame_kernel_grad()
            close(got.reshape(dw.shape), dw, what=f"{name} {i} dW")
            close(gb, db, what=f"{name} {i} db")
    for L, w0 in zip(trained, before):
        want = Adam(w0.size, 1e-4, dtype=np.float32).step(w0, L.grad.cpu().numpy())
        np.testing.assert_allclose(L.canon.cpu().numpy(), want, rtol=0, atol=2e-7)
    first = D.obs_encoder.layers[0]
    kern, _ = first.kernel_bias()
    assert kern.shape == (3, 3, 1, 32) and not first.canon.cpu().numpy()[:16 * 32].reshape(2, 2, 4, 32)[~first._real].any()
    assert int(D.step.item()) == 1


def test_discriminator_learns_to_separate():
    D, H, T = _setup(8, 6, seed=2)
    D.lr = 1e-3
    l0 = D.eval(T["g_img"], T["s_next"], T["s"], T["a"])
    for _ in range(40):
        D.learn(T["g_img"], T["s_next"], T["s"], T["a"])
    assert D.eval(T["g_img"], T["s_next"], T["s"], T["a"]) < 0.7 * l0


def test_state_encoder_reads_the_uint8_ring():
    """The discriminator's step with its states named by replay indices (uint8 ring read in place) equals the
    step on the gathered float32 frames bit for bit."""
    import rl_gan_amd as R
    n, a_dim = 4, 6
    outs = []
    for use_ring in (False, True):
        D, H, T = _setup(n, a_dim, seed=5)
        mem = R.FrameMemory(16, (42, 42, 4), rng=R.MT19937Device(3))
        u8 = torch.from_numpy(np.rint(H["s"] * 255).astype(np.uint8)).cuda()
        mem.store_transitions(u8, u8)
        idx = torch.arange(n, device="cuda")
        s = mem.ref(idx) if use_ring else mem.gather(idx)
        losses = D.learn(T["g_img"], T["s_next"], s, T["a"]).clone()
        outs.append((losses, [o.layer.canon.clone() for o in D._opt]))
    assert torch.equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        assert torch.equal(x, y)


def test_single_transition_calls_of_the_acting_loop():
    """`generator.predict(s, a)` and `discriminator.determine(s, a, g)` as the acting loop calls them
    (Curious_GAN_mario.py:67-71): one state in, NumPy out, equal to row 0 of the batched forms."""
    import rl_gan_amd as R
    D, H, T = _setup(3, 6, seed=7)
    G = R.PixelGenerator(256, 6, 1e-4, D.obs_encoder, rng=np.random.RandomState(0), state_encoder=D.state_encoder)
    g = G.predict(H["s"][0], H["a"][0])
    assert g.shape == (42, 42, 1) and g.dtype == np.float32
    want = G.predict_batch(D.state_encoder.forward(T["s"][:1]), T["a"][:1]).cpu().numpy()[0]
    np.testing.assert_array_equal(g, want)
    r = D.determine(H["s"][0], H["a"][0], g)
    assert r.shape == (1, 1) and 0.0 < r[0, 0] < 1.0
    np.testing.assert_array_equal(r, D.determine_batch(T["s"][:1], T["a"][:1], torch.from_numpy(g[None]).cuda()).cpu().numpy())
    assert D.observe_and_compare(H["s_next"][0], g) == 0.
