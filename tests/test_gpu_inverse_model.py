"""The inverse model's training step on the device (`StateActionPredictor`, Conv_model.py:787-881) against the float64
restatement: logits in both batch-norm modes, the loss, every gradient of the A_pred variables and the state
encoder's (1e-5 relative to the tensor scale, fp32), and the Adam update given those gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n, ac, seed=0):
    import rl_gan_amd as R
    rng = np.random.RandomState(seed)
    f32 = lambda v: np.asarray(v).astype(np.float32).astype(np.float64)
    u = lambda lim, shape: f32(rng.uniform(-lim, lim, shape))
    feat = 9 * 9 * 32                                       # 42 -> 20 -> 9
    st_params = [(u(0.2, (3, 3, 4, 32)), u(0.1, 32)), (u(0.1, (3, 3, 32, 32)), u(0.1, 32))]
    st_dense = (u(0.03, (feat, 256)), f32(np.full(256, 0.001)))
    P = dict(w1_s=u(0.1, (256, 500)), w1_s_=u(0.1, (256, 500)), b1=f32(np.full(500, 0.001)), g1=u(0.5, 500) + 1.0, be1=u(0.5, 500) + 1.0,
             w2=u(0.1, (500, 100)), b2=f32(np.full(100, 0.001)), g2=u(0.5, 100) + 1.0, be2=u(0.5, 100) + 1.0,
             w3=u(0.2, (100, ac)), b3=f32(np.full(ac, 0.001)), g3=u(0.5, ac) + 1.0, be3=u(0.5, ac))
    s = f32(rng.randint(0, 256, (n, 42, 42, 4)) / 255.0)
    s_next = f32(rng.randint(0, 256, (n, 42, 42, 4)) / 255.0)
    a = np.zeros((n, ac))
    a[np.arange(n), rng.randint(0, ac, n)] = 1.0            # one-hot actions (Conv_model.py:790)
    enc = R.ConvEncoder(42, 42, 4, R.universe_head(2), params=st_params, dense=256, dense_params=st_dense)
    M = R.StateActionPredictor(ac, 1e-4, enc, params=P)
    host = dict(s=s, s_next=s_next, a=a, P=P, st_params=st_params, st_dense=st_dense)
    dev = {k: torch.from_numpy(host[k].astype(np.float32)).cuda() for k in ("s", "s_next", "a")}
    return M, host, dev


@pytest.mark.parametrize("n,ac", [(6, 6), (9, 14), (70, 3)])
def test_inverse_model_step_matches_restatement(n, ac):
    from gpu_util import close
    from oracle import inverse_model as om
    from oracle import pixel_gan as op
    from oracle.optim import Adam
    M, H, T = _setup(n, ac)
    P = H["P"]
    lay = op.oc.universe_head(2)
    phis, _ = op.encoder(H["s"], H["st_params"], H["st_dense"], lay)
    phis_n, _ = op.encoder(H["s_next"], H["st_params"], H["st_dense"], lay)
    for training in (False, True):
        want, _ = om.net_forward(phis, phis_n, P, training)
        close(M.logits(T["s"], T["s_next"], training).cpu().numpy(), want, what=f"logits training={training}")
    le = om.eval_loss(H["s"], H["s_next"], H["a"], P, H["st_params"], H["st_dense"])
    assert abs(M.eval(T["s"], T["s_next"], T["a"]) - le) <= 1e-5 * le
    loss, g = om.loss_and_grads(H["s"], H["s_next"], H["a"], P, H["st_params"], H["st_dense"])
    trained = [o.layer for o in M._opt]
    before = [L.canon.cpu().numpy().copy() for L in trained]
    got = float(M.learn(T["s"], T["s_next"], T["a"]).item())
    assert abs(got - loss) <= 1e-5 * loss
    g1 = M.l1.grad.cpu().numpy()
    W = g1[:512 * 500].reshape(512, 500)
    close(W[:256], g["w1_s"], what="dw1_s"); close(W[256:], g["w1_s_"], what="dw1_s_")
    # batch norm removes the column mean, so these bias gradients are exactly 0 and the device leaves rounding noise of
    # the sums behind: the allowance is 1e-5 of the layer's largest kernel gradient
    close(g1[512 * 500:], g["b1"], what="db1", atol=1e-5 * np.abs(g["w1_s"]).max())
    g2 = M.l2.grad.cpu().numpy()
    close(g2[:500 * 100].reshape(500, 100), g["w2"], what="dw2"); close(g2[500 * 100:], g["b2"], what="db2", atol=1e-5 * np.abs(g["w2"]).max())
    g3 = M.l3.grad.cpu().numpy()
    pad = M.ac_pad
    W3 = g3[:100 * pad].reshape(100, pad)
    close(W3[:, :ac], g["w3"], what="dw3"); assert not W3[:, ac:].any()
    close(g3[100 * pad:][:ac], g["b3"], what="db3", atol=1e-5 * np.abs(g["w3"]).max())
    for bn, kg, kb, width in ((M.bn1, "g1", "be1", 500), (M.bn2, "g2", "be2", 100), (M.bn3, "g3", "be3", ac)):
        gb = bn.grad.cpu().numpy()
        close(gb[:bn.n][:width], g[kg], what="d" + kg); close(gb[bn.n:][:width], g[kb], what="d" + kb)
        assert not gb[:bn.n][width:].any() and not gb[bn.n:][width:].any()
    enc = M.state_encoder
    for i, (L, (dw, db)) in enumerate(zip(enc.layers + [enc.head], g["state_encoder"])):
        gl = L.grad.cpu().numpy()
        close(gl[:gl.size - db.size].reshape(dw.shape), dw, what=f"state_encoder {i} dW")
        close(gl[gl.size - db.size:], db, what=f"state_encoder {i} db")
    for L, w0 in zip(trained, before):
        want = Adam(w0.size, 1e-4, dtype=np.float32).step(w0, L.grad.cpu().numpy())
        np.testing.assert_allclose(L.canon.cpu().numpy(), want, rtol=0, atol=2e-7)
    assert int(M.step.item()) == 1


def test_inverse_model_learns_the_actions():
    M, H, T = _setup(12, 6, seed=3)
    M.lr = 1e-3
    l0 = float(M.learn(T["s"], T["s_next"], T["a"]).item())
    for _ in range(60):
        M.learn(T["s"], T["s_next"], T["a"])
    assert float(M.learn(T["s"], T["s_next"], T["a"]).item()) < 0.5 * l0


@pytest.mark.parametrize("m,n", [(7, 100), (256, 500), (1000, 500), (4101, 8), (20000, 100)])
def test_batchnorm_kernels(m, n):
    """Both forms of the batch-norm kernels (one launch up to 256 rows, row chunks above) against the restatement, with a
    column mean far from zero: forward in both modes, and the gradient of the training form."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import inverse_model as om
    rng = np.random.RandomState(m + n)
    f32 = lambda v: np.asarray(v).astype(np.float32).astype(np.float64)
    z = f32(rng.standard_normal((m, n)) * rng.uniform(0.5, 2, n) + rng.uniform(-20, 20, n))
    gamma, beta = f32(rng.uniform(0.5, 1.5, n)), f32(rng.uniform(0, 3, n))
    dy = f32(rng.standard_normal((m, n)))
    bn = R.BatchNorm(n, gamma, beta)
    zt, dyt = torch.from_numpy(z.astype(np.float32)).cuda(), torch.from_numpy(dy.astype(np.float32)).cuda()
    want_inf, _ = om.bn_forward(z, gamma, beta, training=False)
    close(bn.forward(zt, False, True).cpu().numpy(), om.relu6(want_inf), what="inference")
    for relu6 in (False, True):
        want, cache = om.bn_forward(z, gamma, beta, training=True)
        y = bn.forward(zt, True, relu6)
        close(y.cpu().numpy(), om.relu6(want) if relu6 else want, what=f"training relu6={relu6}")
        close(bn.stats.cpu().numpy()[:n], z.mean(0), what="batch mean")
        close(bn.stats.cpu().numpy()[n:], 1 / np.sqrt(z.var(0) + om.EPS), what="1/std")
        d = dy * ((want > 0) & (want < 6)) if relu6 else dy
        dz, dg, db = om.bn_backward(d, cache)
        got = bn.backward(dyt, y, zt, relu6)
        close(bn.grad.cpu().numpy()[:n], dg, what="dgamma", atol=1e-5 * np.abs(d).sum(0).max())
        close(bn.grad.cpu().numpy()[n:], db, what="dbeta", atol=1e-5 * np.abs(d).sum(0).max())
        close(got.cpu().numpy(), dz, what="dz", atol=1e-5 * np.abs(dz).max())


def test_batchnorm_properties_at_a_large_batch():
    """65,536 rows x 500 columns, too many for the float64 restatement to be worth its time here: the output of the
    training form has column mean beta and column deviation gamma * sqrt(var / (var + eps)), inference leaves the
    statistics alone, and the input gradient sums to zero over every column."""
    import rl_gan_amd as R
    m, n = 65536, 500
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    z = torch.randn(m, n, device="cuda", generator=g) * 1.7 + 11.0
    gamma = np.linspace(0.5, 1.5, n); beta = np.linspace(-1, 1, n)
    bn = R.BatchNorm(n, gamma, beta)
    y = bn.forward(z, True, False)
    z64 = z.double()
    var = z64.var(0, unbiased=False)
    np.testing.assert_allclose(y.double().mean(0).cpu().numpy(), beta, atol=2e-5)
    np.testing.assert_allclose(y.double().std(0, unbiased=False).cpu().numpy(), gamma * np.sqrt((var / (var + 1e-3)).cpu().numpy()), rtol=2e-5)
    np.testing.assert_allclose(bn.stats[:n].cpu().numpy(), z64.mean(0).cpu().numpy(), rtol=1e-6)
    dy = torch.randn(m, n, device="cuda", generator=g)
    dz = bn.backward(dy, y, z, False)
    scale = float(dz.abs().sum(0).max())
    assert float(dz.double().sum(0).abs().max()) <= 1e-5 * scale
    np.testing.assert_allclose(bn.grad[n:].cpu().numpy(), dy.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-5 * float(dy.abs().sum(0).max()))
    yi = bn.forward(z, False, False)
    want = torch.from_numpy(gamma).cuda() * z64 / np.sqrt(1 + 1e-3) + torch.from_numpy(beta).cuda()
    assert float((yi.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
