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


@pytest.mark.parametrize("m,k,n", [(37, 64, 256), (130, 2592, 256), (64, 12800, 256), (5, 20, 12)])
def test_dense_head_forward_and_backward(m, k, n):
    """The 256-unit ELU head (Conv_model.py:24, :52-56) on 64 (doom), 2,592 (42x42 universe) and 12,800
    (84x84 universe, two layers) features, ragged batch sizes, plus a shape smaller than one tile."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    rng = np.random.RandomState(m + k)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    x = f32(rng.standard_normal((m, k)))
    w = f32(rng.uniform(-1, 1, (k, n)) * np.sqrt(6.0 / (k + n)))
    b = f32(rng.uniform(-0.1, 0.1, n))
    y, cache = oc.dense_forward(x, w, b)
    dy = f32(rng.standard_normal(y.shape))
    dx, dw, db = oc.dense_backward(dy, w, cache)
    L = R.DenseLayer(k, n, kernel=w, bias=b)
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    yt = L.forward(xt)
    close(yt.cpu().numpy(), y, what="dense y")
    dxt = L.backward(xt, yt, torch.from_numpy(dy.astype(np.float32)).cuda())
    close(dxt.cpu().numpy(), dx, what="dense dx")
    g = L.grad.cpu().numpy()
    close(g[:k * n].reshape(k, n), dw, what="dense dW")
    close(g[k * n:], db, what="dense db")
    zt = L.forward(xt, act=False)
    close(zt.cpu().numpy(), x @ w + b, what="dense linear")


@pytest.mark.parametrize("m,k,n,ns", [(32, 200, 1152, 9), (288, 128, 1024, 8), (5, 20, 12, 3)])
def test_dense_input_gradient_in_slabs(m, k, n, ns):
    """dX = dZ * W^T with the sum over the n hidden units cut into slabs (the batch-32 shapes of the pixel generator's
    decoder, where the output alone has a handful of tiles): equal to the float64 product, and the same bits on a second
    call (slabs are added in slab order)."""
    from rl_gan_amd import _lib
    from gpu_util import close
    rng = np.random.RandomState(m + n)
    dz = rng.standard_normal((m, n)).astype(np.float32)
    w = (rng.uniform(-1, 1, (k, n)) * np.sqrt(6.0 / (k + n))).astype(np.float32)
    wb = torch.from_numpy(np.concatenate([w.ravel(), np.zeros(n, np.float32)])).cuda()
    dzt = torch.from_numpy(dz).cuda()
    slabs = torch.empty(ns * m * k, device="cuda")
    outs = []
    for _ in range(2):
        dx = torch.full((m, k), float("nan"), device="cuda")
        assert _lib.lib.rlgan_dense_backward_input_split(dzt.data_ptr(), n, wb.data_ptr(), dx.data_ptr(), k, m, k, n,
                                                         slabs.data_ptr(), ns, None) == 0
        outs.append(dx.cpu().numpy())
    close(outs[0], dz.astype(np.float64) @ w.astype(np.float64).T, what="dX in slabs")
    np.testing.assert_array_equal(outs[0], outs[1])
    one = torch.empty((m, k), device="cuda")
    assert _lib.lib.rlgan_dense_backward_input(dzt.data_ptr(), n, wb.data_ptr(), one.data_ptr(), k, m, k, n, None) == 0
    close(one.cpu().numpy(), outs[0].astype(np.float64), what="one slab against several")
    assert _lib.lib.rlgan_dense_backward_input_split(dzt.data_ptr(), n, wb.data_ptr(), one.data_ptr(), k, m, k, n, None, 4, None) == -1


@pytest.mark.parametrize("m,k,n,ns", [(2304, 128, 1024, 16), (300, 200, 1152, 1), (40, 20, 12, 3)])
def test_dense_kernel_gradient_without_the_bias_row(m, k, n, ns):
    """gw[k, n] = x^T dz, the weight gradient of a layer whose bias gradient comes from elsewhere (the transposed
    convolutions in their cols form): equal to the float64 product, equal to the first k rows of
    rlgan_dense_backward_weights with the same slabs (same products in the same order), and nothing is written past row k."""
    from rl_gan_amd import _lib
    from gpu_util import close
    rng = np.random.RandomState(k + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    dz = rng.standard_normal((m, n)).astype(np.float32)
    xt, dzt = torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda()
    slabs = torch.empty(max(1, ns) * (k + 1) * n, device="cuda")
    gw = torch.full(((k + 1), n), float("nan"), device="cuda")
    assert _lib.lib.rlgan_dense_backward_kernel(xt.data_ptr(), k, dzt.data_ptr(), n, gw.data_ptr(), m, k, n, slabs.data_ptr(), ns, None) == 0
    got = gw.cpu().numpy()
    assert np.isnan(got[k]).all()                                            # the row a bias gradient would take is untouched
    close(got[:k], x.astype(np.float64).T @ dz.astype(np.float64), what="kernel gradient")
    gwb = torch.empty(((k + 1), n), device="cuda")
    assert _lib.lib.rlgan_dense_backward_weights(xt.data_ptr(), k, dzt.data_ptr(), n, gwb.data_ptr(), m, k, n, slabs.data_ptr(), ns, None) == 0
    np.testing.assert_array_equal(got[:k], gwb.cpu().numpy()[:k])
    close(gwb.cpu().numpy()[k], dz.astype(np.float64).sum(0), what="bias row of the full form")
    assert _lib.lib.rlgan_dense_backward_kernel(xt.data_ptr(), k - 1, dzt.data_ptr(), n, gw.data_ptr(), m, k, n, None, 1, None) == -1


@pytest.mark.parametrize("m,k,ldx,n", [(70, 18, 20, 12), (130, 201, 204, 72), (64, 66, 68, 500)])
def test_dense_products_with_a_width_that_is_not_a_multiple_of_four(m, k, ldx, n):
    """Rows of x padded to a multiple of four floats with k itself odd-sized: border tiles are then read element by
    element instead of in aligned groups of four; forward and weight gradient against float64, padding never read as data."""
    from rl_gan_amd import _lib
    from gpu_util import close
    rng = np.random.RandomState(k + n)
    x = np.full((m, ldx), np.nan, np.float32)
    x[:, :k] = rng.standard_normal((m, k))
    wb = (rng.uniform(-1, 1, (k + 1, n)) * 0.2).astype(np.float32)
    dz = rng.standard_normal((m, n)).astype(np.float32)
    xt, wbt, dzt = (torch.from_numpy(a).cuda() for a in (x, wb, dz))
    y = torch.empty((m, n), device="cuda")
    assert _lib.lib.rlgan_dense_forward(xt.data_ptr(), ldx, wbt.data_ptr(), y.data_ptr(), n, m, k, n, 0, None, 1, None) == 0
    x64 = x[:, :k].astype(np.float64)
    close(y.cpu().numpy(), x64 @ wb[:k].astype(np.float64) + wb[k].astype(np.float64), what="forward, odd k")
    gwb = torch.empty((k + 1, n), device="cuda")
    slabs = torch.empty(2 * (k + 1) * n, device="cuda")
    assert _lib.lib.rlgan_dense_backward_weights(xt.data_ptr(), ldx, dzt.data_ptr(), n, gwb.data_ptr(), m, k, n, slabs.data_ptr(), 2, None) == 0
    g = gwb.cpu().numpy()
    close(g[:k], x64.T @ dz.astype(np.float64), what="dW, odd k")
    close(g[k], dz.astype(np.float64).sum(0), what="db, odd k")


@pytest.mark.parametrize("n", [6, 200])
def test_dz_handed_from_the_layer_above(n):
    """A layer's input gradient multiplied by ELU'(its input) inside the kernel is the dZ of the layer below: the same bits
    as the product formed afterwards, and the layer below, given that dZ, leaves the gradients it computes from (dY, y)
    (`universe_feature_abstraction`, Conv_model.py:36-49: 42x42x4 -> 20x20x32 -> 9x9x32)."""
    import rl_gan_amd as R
    from rl_gan_amd import _lib
    rng = np.random.RandomState(n)
    L1 = R.ConvLayer(42, 42, 4, 32, 3, 2, rng=rng, bias_init=0.01)
    L2 = R.ConvLayer(20, 20, 32, 32, 3, 2, rng=rng, bias_init=0.01)
    x = torch.from_numpy(rng.standard_normal((n, 42, 42, 4)).astype(np.float32)).cuda()
    y1 = L1.forward(x)
    y2 = L2.forward(y1)
    dy2 = torch.from_numpy(rng.standard_normal(tuple(y2.shape)).astype(np.float32)).cuda()
    dy1 = L2.backward(y1, y2, dy2)
    dz1 = torch.empty_like(dy1)
    assert _lib.lib.rlgan_elu_backward(dy1.data_ptr(), y1.data_ptr(), dz1.data_ptr(), dy1.numel(), None) == 0
    got = L2.backward(y1, y2, dy2, dx_elu=True)
    assert torch.equal(got, dz1)
    dx_ref = L1.backward(x, y1, dy1)
    g_ref = L1.grad.clone()
    dx = L1.backward(x, y1, dz1, dy_is_dz=True)
    assert torch.equal(L1.grad, g_ref) and torch.equal(dx, dx_ref)


@pytest.mark.parametrize("h,cin,cout,k,st,n", [(12, 8, 16, 3, 3, 5), (9, 16, 32, 3, 2, 3), (10, 4, 4, 2, 4, 6), (41, 32, 32, 3, 2, 70)])
def test_input_gradient_times_elu_prime_of_the_input_on_every_kernel(h, cin, cout, k, st, n):
    """rlgan_conv_backward_input_dz on shapes that take the other input-gradient kernels (per-pixel-group, four-channel,
    few images): dx * ELU'(x), the bits of the separate pass, whichever kernel runs and whether it applies the factor
    itself or an elementwise pass follows."""
    import rl_gan_amd as R
    from rl_gan_amd import _lib
    rng = np.random.RandomState(h + cin)
    L = R.ConvLayer(h, h, cin, cout, k, st, rng=rng, bias_init=0.01)
    x = torch.from_numpy(rng.standard_normal((n, h, h, cin)).astype(np.float32)).cuda()
    y = L.forward(x)
    dy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
    dx = L.backward(x, y, dy, need_dw=False)
    want = torch.empty_like(dx)
    assert _lib.lib.rlgan_elu_backward(dx.data_ptr(), x.data_ptr(), want.data_ptr(), dx.numel(), None) == 0
    got = L.backward(x, y, dy, need_dw=False, dx_elu=True)
    assert torch.equal(got, want)


def test_encoder_with_dense_head_end_to_end():
    """doom head + fc(256): features, and the gradient reaching the first conv layer."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    rng = np.random.RandomState(11)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    n = 6
    x = f32(rng.uniform(0, 1, (n, 84, 84, 4)))
    params = [(f32(w), f32(b)) for w, b in oc.init_encoder(rng, oc.DOOM_HEAD, 4, bias=0.001)]
    wd, bd = f32(rng.uniform(-1, 1, (64, 256)) * np.sqrt(6.0 / 320)), f32(np.full(256, 0.001))
    feat, caches = oc.encoder_forward(x, params, oc.DOOM_HEAD)
    y, dc = oc.dense_forward(feat.reshape(n, -1), wd, bd)
    dy = f32(rng.standard_normal(y.shape))
    dfeat, dw, db = oc.dense_backward(dy, wd, dc)
    _, grads = oc.encoder_backward(dfeat.reshape(feat.shape), params, oc.DOOM_HEAD, caches)
    enc = R.ConvEncoder(84, 84, 4, oc.DOOM_HEAD, params=params, dense=256, dense_params=(wd, bd))
    out = enc.forward(torch.from_numpy(x.astype(np.float32)).cuda())
    assert out.shape == (n, 256) and enc.features == 256 and enc.conv_features == 64
    close(out.cpu().numpy(), y, what="fc output")
    enc.backward(torch.from_numpy(dy.astype(np.float32)).cuda())
    close(enc.head.grad.cpu().numpy()[:64 * 256].reshape(64, 256), dw, what="fc dW")
    g0 = enc.layers[0].grad.cpu().numpy()
    close(g0[:grads[0][0].size].reshape(grads[0][0].shape), grads[0][0], what="conv l1 dW through the head")


def test_dense_argument_errors():
    from rl_gan_amd import _lib
    x = torch.zeros(8, 20, device="cuda"); wb = torch.zeros(21 * 12, device="cuda"); y = torch.zeros(8, 12, device="cuda")
    f = _lib.lib.rlgan_dense_forward
    assert f(x.data_ptr(), 20, wb.data_ptr(), y.data_ptr(), 12, 8, 20, 12, 1, None, 1, None) == 0
    assert f(x.data_ptr(), 18, wb.data_ptr(), y.data_ptr(), 12, 8, 20, 12, 1, None, 1, None) == -1      # ldx < k
    assert f(x.data_ptr(), 22, wb.data_ptr(), y.data_ptr(), 12, 8, 20, 12, 1, None, 1, None) == -5      # ldx not a multiple of 4
    assert f(x.data_ptr() + 4, 20, wb.data_ptr(), y.data_ptr(), 12, 8, 20, 12, 1, None, 1, None) == -5  # base not 16-byte aligned
    assert f(None, 20, wb.data_ptr(), y.data_ptr(), 12, 8, 20, 12, 1, None, 1, None) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("h,cin,cout,k,st", [(84, 4, 8, 5, 4), (41, 32, 32, 3, 2), (9, 16, 32, 3, 2), (12, 8, 16, 3, 3), (10, 4, 4, 2, 4)])
def test_input_gradient_two_ways(h, cin, cout, k, st):
    """The stride-phase MFMA form of dX against the gather from the canonical kernel, including a layer
    whose stride exceeds its kernel (pixels no patch covers get a zero gradient).  With four input channels both
    calls run the one-thread-per-pixel kernel: once with ELU' folded into its reads, once on a gradient masked
    beforehand (that kernel against the float64 restatement: test_encoder_forward_and_backward)."""
    import rl_gan_amd as R
    from gpu_util import close
    rng = np.random.RandomState(h + k)
    L = R.ConvLayer(h, h, cin, cout, k, st, rng=rng, bias_init=0.01)
    assert L.wdx is not None
    x = torch.rand(3, h, h, cin, device="cuda")
    y = L.forward(x)
    dy = torch.randn_like(y)
    a = L.backward(x, y, dy).clone()
    L.gather_dx = True
    b = L.backward(x, y, dy)
    close(a.cpu().numpy(), b.cpu().numpy().astype(np.float64), what="dX mfma vs gather")


@pytest.mark.parametrize("n", [3, 70, 600])
def test_input_gradient_of_the_3x3_layer_at_any_batch(n):
    """dX of the 32 -> 32 channel 3x3 stride-2 layer on 20x20 maps (Conv_model.py:38-44) takes the per-pixel-group kernel
    below 64 images and the image-in-LDS kernel above, with one to four images per workgroup depending on how many
    workgroups that leaves: every form against the float64 restatement, and an image's gradient does not depend on the
    batch it came in beyond the kernel choice (rows 0..2 of the 70- and 600-image batches are the same bits)."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    rng = np.random.RandomState(5)
    L = R.ConvLayer(20, 20, 32, 32, 3, 2, rng=rng, bias_init=0.01)
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.rand(600, 20, 20, 32, device="cuda", generator=g)[:n].contiguous()
    y = L.forward(x)
    dy = torch.randn(600, *y.shape[1:], device="cuda", generator=g)[:n].contiguous()
    dx = L.backward(x, y, dy, need_dw=False)
    w, b = L.kernel_bias()
    m = min(n, 3)
    _, cache = oc.conv_forward(x[:m].cpu().numpy().astype(np.float64), w.astype(np.float64), b.astype(np.float64), 3, 2)
    dxo, _, _ = oc.conv_backward(dy[:m].cpu().numpy().astype(np.float64), w.astype(np.float64), cache, 3, 2)
    close(dx[:m].cpu().numpy(), dxo, what=f"dX of the first images of {n}")
    if n == 600:
        d70 = L.backward(x[:70].contiguous(), y[:70].contiguous(), dy[:70].contiguous(), need_dw=False)
        assert torch.equal(d70[:3], dx[:3])


@pytest.mark.parametrize("h,cin,cout,k,st,n", [(84, 4, 8, 5, 4, 7), (42, 4, 32, 3, 2, 9), (20, 8, 16, 3, 2, 40), (42, 4, 32, 3, 2, 30),
                                               (21, 4, 16, 3, 2, 11), (21, 4, 32, 2, 1, 30)])
def test_weight_gradient_two_ways(h, cin, cout, k, st, n):
    """Layers with few patch elements have two weight-gradient kernels (one slab per workgroup, or per wave when
    every slab covers a long run of pixels): both against the float64 restatement, on float32 frames and on the
    uint8 ring.  The last case is the block form of a one-channel first layer (2x2 taps, stride 1, 21x21 blocks)."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    rng = np.random.RandomState(h + cout)
    L = R.ConvLayer(h, h, cin, cout, k, st, rng=rng, bias_init=0.01)
    u8 = torch.randint(0, 256, (n, h * h * cin), dtype=torch.uint8, device="cuda")
    x = (u8.to(torch.float32) * torch.tensor(np.float32(1.0 / 255), device="cuda")).reshape(n, h, h, cin)
    y = L.forward(x)
    dy = torch.randn_like(y)
    w, b = L.kernel_bias()
    w64 = w.astype(np.float64)
    yo, cache = oc.conv_forward(x.cpu().numpy().astype(np.float64), w64, b.astype(np.float64), k, st)
    _, dw, db = oc.conv_backward(dy.cpu().numpy().astype(np.float64), w64, cache, k, st)
    want = np.concatenate([dw.ravel(), db.ravel()])
    m = n * L.h_out * L.w_out
    got = {}
    # pixels per slab: few (workgroup kernel) / at least 256 (wave kernel); a 3x3 first layer on small frames runs the
    # image-in-LDS kernel either way, one image per workgroup, and with 4 slabs several images per workgroup (the next
    # image is fetched while the current one is multiplied)
    for ns in (1024, max(4, m // 300)) + ((4,) if cin == 4 and k <= 3 else ()):
        L.n_slabs, L._small = ns, False
        L.grad.zero_()
        L.backward(x, y, dy, need_dx=False)
        got[ns] = L.grad.clone()
        close(got[ns].cpu().numpy(), want, what=f"dW with {ns} slabs")
        L.grad.zero_()
        L.backward_frames(R.FrameRef(u8, None, n=n), y, dy)
        assert torch.equal(L.grad, got[ns])


@pytest.mark.parametrize("n,ns", [(1024, 1024), (1100, 1024), (1500, 100)])
def test_weight_gradient_of_the_32_channel_layer_with_images_in_lds(n, ns):
    """From 1024 images on, the weight gradient of the 3x3 stride-2 layer with 32 input channels on 20x20 maps keeps whole
    images in two LDS buffers (four waves multiply one while four fill the other): against the float64 restatement, with
    as many workgroups as images allow (one to five images each) and with 100 slabs (15 images each, the last fewer), with
    and without the ELU derivative, and the same bits on a second call."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    rng = np.random.RandomState(n)
    L = R.ConvLayer(20, 20, 32, 32, 3, 2, rng=rng, bias_init=0.01)
    x = torch.randn((n, 20, 20, 32), device="cuda")
    y = L.forward(x)
    dy = torch.randn_like(y)
    w, b = L.kernel_bias()
    w64 = w.astype(np.float64)
    _, cache = oc.conv_forward(x.cpu().numpy().astype(np.float64), w64, b.astype(np.float64), 3, 2)
    _, dw, db = oc.conv_backward(dy.cpu().numpy().astype(np.float64), w64, cache, 3, 2)
    L.n_slabs, L._slabs = ns, None
    L.grad.zero_()
    L.backward(x, y, dy, need_dx=False)
    first = L.grad.clone()
    close(first.cpu().numpy(), np.concatenate([dw.ravel(), db.ravel()]), what="dW, images in LDS")
    L.grad.zero_()
    L.backward(x, y, dy, need_dx=False)
    assert torch.equal(L.grad, first)
    few = L.grad.clone()                                   # fewer than 1024 images take the other kernel: same sums, another order
    L.grad.zero_()
    L.backward(x[:1000].contiguous(), y[:1000].contiguous(), dy[:1000].contiguous(), need_dx=False)
    assert not torch.equal(L.grad, few)
    L.grad.zero_()
    L.backward(x, y, dy, need_dx=False, act=False)         # no activation: dZ = dY
    yl, cache = oc.conv_forward(x.cpu().numpy().astype(np.float64), w64, b.astype(np.float64), 3, 2, act=False)
    _, dw, db = oc.conv_backward(dy.cpu().numpy().astype(np.float64), w64, cache, 3, 2, act=False)
    close(L.grad.cpu().numpy(), np.concatenate([dw.ravel(), db.ravel()]), what="dW without the activation")


@pytest.mark.parametrize("name", ["doom", "universe"])
def test_first_layer_streamed_in_bands(name):
    """The first layer of the 84x84x4 encoders (5x5 stride 4 -> 8 channels, Conv_model.py:15-16; 3x3 stride 2 -> 32
    channels, Conv_model.py:38-39) walks (image, band of output rows) items with persistent workgroups.  300 images are
    more items than workgroups, batches of 7 fewer: both give the same bits per image, from float32 frames and from the
    uint8 ring through an index array (slots out of range read as zero images), and the float64 restatement's values."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    f, k, st = (8, 5, 4) if name == "doom" else (32, 3, 2)
    n = 300
    L = R.ConvLayer(84, 84, 4, f, k, st, rng=np.random.RandomState(7), bias_init=0.01)
    g = torch.Generator(device="cuda").manual_seed(2)
    ring = torch.randint(0, 256, (n + 2, 84 * 84 * 4), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randperm(n + 2, device="cuda", generator=g)[:n].contiguous()
    idx[5], idx[n - 1] = -1, n + 9                                   # no such slots
    ok = ((idx >= 0) & (idx < n + 2)).to(torch.float32).reshape(n, 1)
    x = (ring[idx.clamp(0, n + 1)].to(torch.float32) * torch.tensor(np.float32(1.0 / 255), device="cuda") * ok).reshape(n, 84, 84, 4).contiguous()
    y = L.forward(x)
    for i in range(0, n, 7 * 6):
        assert torch.equal(L.forward(x[i:i + 7].contiguous()), y[i:i + 7])
    assert torch.equal(L.forward_frames(R.FrameRef(ring, idx, 1.0 / 255)), y)
    w, b = L.kernel_bias()
    yo, _ = oc.conv_forward(x[3:7].cpu().numpy().astype(np.float64), w.astype(np.float64), b.astype(np.float64), k, st)
    close(y[3:7].cpu().numpy(), yo, what=f"{name} first layer")
    z = L.forward(x[:4].contiguous(), act=False)
    zo, _ = oc.conv_forward(x[:4].cpu().numpy().astype(np.float64), w.astype(np.float64), b.astype(np.float64), k, st, act=False)
    close(z.cpu().numpy(), zo, what=f"{name} first layer, linear")


@pytest.mark.parametrize("hw,n", [(42, 7), (12, 3)])
def test_single_frame_first_layer_on_2x2_blocks(hw, n):
    """One-channel frames (the single-frame encoder, Curious_GAN_mario.py:313-314): the 3x3 stride-2 first layer runs as a
    2x2 stride-1 convolution of the frame's 2x2 blocks.  Block regrouping and its inverse, features, the frames' gradient
    and every kernel / bias gradient in the reference's shapes against the float64 restatement on the plain frames; the
    seven kernel entries outside the 3x3 window get no gradient."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    rng = np.random.RandomState(hw)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    spec = oc.universe_head(2)
    x = f32(rng.uniform(0, 1, (n, hw, hw, 1)))
    params = [(f32(w), f32(b)) for w, b in oc.init_encoder(rng, spec, 1, bias=0.01)]
    y, caches = oc.encoder_forward(x, params, spec)
    dy = f32(rng.standard_normal(y.shape))
    dx, grads = oc.encoder_backward(dy, params, spec, caches)
    enc = R.ConvEncoder(hw, hw, 1, spec, params=params)
    first = enc.layers[0]
    assert isinstance(first, R.conv.SingleFrameConv) and (first.k, first.stride, first.c_in) == (2, 1, 4)
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    blocks = first.regroup(xt)
    want = x.reshape(n, hw // 2, 2, hw // 2, 2).transpose(0, 1, 3, 2, 4).reshape(n, hw // 2, hw // 2, 4)
    np.testing.assert_array_equal(blocks.cpu().numpy(), want.astype(np.float32))
    assert torch.equal(first.ungroup(blocks), xt)
    kern, bias = first.kernel_bias()
    np.testing.assert_array_equal(kern, params[0][0].astype(np.float32)); np.testing.assert_array_equal(bias, params[0][1].astype(np.float32))
    feat = enc.forward(xt)
    close(feat.cpu().numpy(), y.reshape(n, -1), what="features")
    dxt = enc.backward(torch.from_numpy(dy.reshape(n, -1).astype(np.float32)).cuda(), need_dx=True)
    assert dxt.shape == xt.shape
    close(dxt.cpu().numpy(), dx, what="d frames")
    dw0, db0 = first.frame_kernel_grad()                    # asserts that the entries outside the window are zero
    close(dw0, grads[0][0], what="layer 0 dW"); close(db0, grads[0][1], what="layer 0 db")
    g1 = enc.layers[1].grad.cpu().numpy()
    close(g1[:grads[1][0].size].reshape(grads[1][0].shape), grads[1][0], what="layer 1 dW")
    # the input gradient alone leaves the weight gradients as they are
    keep = [L.grad.clone() for L in enc.layers]
    enc.forward(xt)
    again = enc.backward(torch.from_numpy(dy.reshape(n, -1).astype(np.float32)).cuda(), need_dx=True, need_dw=False)
    assert torch.equal(again, dxt) and all(torch.equal(a, L.grad) for a, L in zip(keep, enc.layers))


def test_blocks2x2_argument_errors():
    from rl_gan_amd._lib import lib
    a = torch.zeros(4 * 6 * 6 + 8, device="cuda"); b = torch.zeros(4 * 6 * 6 + 8, device="cuda")
    f = lib.rlgan_blocks2x2
    assert f(a.data_ptr(), b.data_ptr(), 4, 6, 6, 1, None) == 0
    assert f(a.data_ptr(), b.data_ptr(), 4, 5, 6, 1, None) == -2          # odd height
    assert f(a.data_ptr(), b.data_ptr(), 4, 6, 7, 0, None) == -2
    assert f(a.data_ptr() + 8, b.data_ptr(), 4, 6, 6, 1, None) == -5      # not 16-byte aligned
    assert f(None, b.data_ptr(), 4, 6, 6, 1, None) == -1 and f(a.data_ptr(), b.data_ptr(), 0, 6, 6, 1, None) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("c_in", [1, 4])
def test_one_batch_encoded_in_two_calls(c_in):
    """`forward_rows` into an activation set, half a batch at a time, then `use` + `backward`: features, the input gradient
    and every weight gradient equal those of one `forward` over the concatenated batch bit for bit (a frame's results do
    not depend on its batch; the gradient sums run over the same rows in the same order)."""
    import rl_gan_amd as R
    rng = np.random.RandomState(c_in)
    enc = R.ConvEncoder(42, 42, c_in, R.universe_head(2), rng=rng, dense=256, bias_init=0.001)
    a, b = torch.rand(5, 42, 42, c_in, device="cuda"), torch.rand(3, 42, 42, c_in, device="cuda")
    d = torch.randn(8, 256, device="cuda")
    want = enc.forward(torch.cat([a, b], 0)).clone()
    dx_want = enc.backward(d, need_dx=True).clone()
    g_want = [L.grad.clone() for L in enc.layers + [enc.head]]
    for L in enc.layers + [enc.head]:
        L.grad.zero_()
    acts = enc.new_activation_set(8)
    fb = enc.forward_rows(b, acts, 5)                      # the second part first: the order of the calls does not matter
    enc.forward(torch.rand(2, 42, 42, c_in, device="cuda"))    # another pass in between leaves the set alone
    fa = enc.forward_rows(a, acts, 0)
    assert torch.equal(fa, want[:5]) and torch.equal(fb, want[5:])
    assert torch.equal(enc.use(acts), want)
    assert torch.equal(enc.backward(d, need_dx=True), dx_want)
    for L, g in zip(enc.layers + [enc.head], g_want):
        assert torch.equal(L.grad, g)


@pytest.mark.parametrize("h,cin,cout,n_few,n_many", [(4, 32, 64, 40, 17000), (4, 32, 32, 300, 16500), (9, 32, 32, 100, 1100)])
def test_last_layers_with_few_tiles(h, cin, cout, n_few, n_many):
    """A 3x3 stride-2 layer on a small map: with at most 16384 output pixels in the batch every wave takes one tile and
    loads all it needs at once, above that the tiles are walked by resident workgroups.  Both against the float64
    restatement, and an image's output is the same bits in either batch (same products in the same order)."""
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import conv as oc
    rng = np.random.RandomState(h + cout)
    spec = ((cout, 3, 2),)
    params = [(w.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64))
              for w, b in oc.init_encoder(rng, spec, cin, bias=0.001)]
    L = R.ConvLayer(h, h, cin, cout, 3, 2, kernel=params[0][0], bias=params[0][1])
    assert n_few * L.h_out * L.w_out <= 16384 < n_many * L.h_out * L.w_out
    x = rng.uniform(-1, 1, (n_many, h, h, cin)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    many = L.forward(xt)
    few = L.forward(xt[:n_few].contiguous())
    assert torch.equal(few, many[:n_few])
    y, _ = oc.encoder_forward(x[:n_few].astype(np.float64), params, spec)
    close(few.cpu().numpy(), y, what="few tiles")
    tail = slice(n_many - 5, n_many)
    y, _ = oc.encoder_forward(x[tail].astype(np.float64), params, spec)
    close(many[tail].cpu().numpy(), y, what="many tiles")
    one = L.forward(xt[7:8].contiguous(), act=False)                        # a lone image, no activation
    assert torch.equal(one, L.forward(xt[:n_few].contiguous(), act=False)[7:8])


def test_elu_absolute_accuracy():
    """tf.nn.elu on the device is exp(z) - 1 through the hardware's exp2, the form TensorFlow's own kernel has: within
    1.5e-7 of the float64 value for every z (about an ulp of the exponential), over twelve decades of |z|."""
    import rl_gan_amd as R
    z = np.concatenate([-np.logspace(-9, np.log10(30.0), 4000), np.linspace(-2, 2, 1001), [0.0, -0.0, -88.0, -200.0]]).astype(np.float32)
    n = 64
    z = np.resize(z, (z.size + n - 1) // n * n).reshape(-1, n)
    L = R.DenseLayer(n, n, kernel=np.eye(n), bias=np.zeros(n))
    y = L.forward(torch.from_numpy(z).cuda()).cpu().numpy().astype(np.float64)
    z64 = z.astype(np.float64)
    want = np.where(z64 > 0, z64, np.expm1(z64))
    assert np.abs(y - want).max() <= 1.5e-7
    assert (y[z > 0] == z[z > 0]).all() and (y >= -1.0).all()
