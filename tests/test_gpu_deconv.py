"""Decoder of the pixel generator on the device (Conv_model.py:457-473) against the float64 restatement:
image, code gradient and every kernel/bias gradient, 1e-5 relative to the tensor scale (fp32)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,ci,co,k,s,act,n", [(1, 200, 128, 3, 2, True, 5), (3, 128, 64, 4, 2, True, 3), (8, 64, 32, 5, 2, True, 4),
                                                (19, 32, 1, 6, 2, False, 3), (4, 8, 6, 2, 3, True, 2),
                                                # the two shapes with fused kernels: odd and larger image counts (two images per
                                                # workgroup forward, chunks of images backward), with and without ReLU
                                                (8, 64, 32, 5, 2, True, 1), (8, 64, 32, 5, 2, False, 7), (8, 64, 32, 5, 2, True, 150),
                                                (19, 32, 1, 6, 2, True, 1), (19, 32, 1, 6, 2, False, 70),
                                                # forward fused only, sixteen images per workgroup
                                                (3, 128, 64, 4, 2, True, 16), (3, 128, 64, 4, 2, False, 45)])
def test_layer_forward_and_backward(h, ci, co, k, s, act, n):
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import deconv as od
    rng = np.random.RandomState(h + k)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    x = f32(rng.standard_normal((n, h, h, ci)))
    w = f32(rng.standard_normal((k, k, co, ci)) * 0.2)
    b = f32(rng.standard_normal(co) * 0.1)
    y, cache = od.deconv_forward(x, w, b, s, act)
    dy = f32(rng.standard_normal(y.shape))
    dx, dw, db = od.deconv_backward(dy, w, cache, s, act)
    L = R.DeconvLayer(h, h, ci, co, k, s, kernel=w, bias=b)
    assert (L.h_out, L.w_out) == y.shape[1:3]
    xt = torch.from_numpy(x.astype(np.float32)).cuda()
    yt = L.forward(xt, act=act)
    close(yt.cpu().numpy(), y, what="y")
    dxt = L.backward(xt, yt, torch.from_numpy(dy.astype(np.float32)).cuda(), act=act)
    close(dxt.cpu().numpy(), dx, what="dx")
    g = L.grad.cpu().numpy()
    close(g[:dw.size].reshape(dw.shape), dw, what="dW")
    close(g[dw.size:], db, what="db")
    # the same bits on a second run (fixed summation orders)
    g1 = L.grad.clone()
    y2 = L.forward(xt, act=act)
    L.backward(xt, y2, torch.from_numpy(dy.astype(np.float32)).cuda(), act=act)
    assert torch.equal(y2, yt) and torch.equal(L.grad, g1)
    # without the input gradient the parameter gradients are the same
    assert L.backward(xt, yt, torch.from_numpy(dy.astype(np.float32)).cuda(), need_dx=False, act=act) is None
    close(L.grad.cpu().numpy(), g1.cpu().numpy(), what="grad without dx")
    if L.fused:
        # the cols form of the same layer agrees to rounding
        L2 = R.DeconvLayer(h, h, ci, co, k, s, kernel=w, bias=b, fused=False)
        assert not L2.fused
        y3 = L2.forward(xt, act=act)
        close(yt.cpu().numpy(), y3.cpu().numpy().astype(np.float64), what="y, fused against cols")
        dx3 = L2.backward(xt, y3, torch.from_numpy(dy.astype(np.float32)).cuda(), act=act)
        close(dxt.cpu().numpy(), dx3.cpu().numpy().astype(np.float64), what="dx, fused against cols")


@pytest.mark.parametrize("act,n", [(True, 64), (False, 128)])
def test_backward_products_through_the_cols_view(act, n):
    """3x3x128 -> 8x8x64 (dec_deconv2, Conv_model.py:463) with a batch whose extents are whole tiles: the backward products
    read dz in place of the gathered cols matrix and leave the bits of the cols form; against float64 as well."""
    import rl_gan_amd as R
    from rl_gan_amd import _lib
    from gpu_util import close
    from oracle import deconv as od
    rng = np.random.RandomState(n)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    x = f32(rng.standard_normal((n, 3, 3, 128)))
    w = f32(rng.standard_normal((4, 4, 64, 128)) * 0.1)
    b = f32(rng.standard_normal(64) * 0.1)
    y, cache = od.deconv_forward(x, w, b, 2, act)
    dy = f32(rng.standard_normal(y.shape))
    dx, dw, db = od.deconv_backward(dy, w, cache, 2, act)
    L = R.DeconvLayer(3, 3, 128, 64, 4, 2, kernel=w, bias=b)
    assert _lib.lib.rlgan_deconv_view_ok(L.spec, n) == 1 and _lib.lib.rlgan_deconv_view_ok(L.spec, n + 1) == 0
    xt, dyt = torch.from_numpy(x.astype(np.float32)).cuda(), torch.from_numpy(dy.astype(np.float32)).cuda()
    yt = L.forward(xt, act=act)
    dx_view = L.backward(xt, yt, dyt, act=act)
    g_view = L.grad.clone()
    close(dx_view.cpu().numpy(), dx, what="dx through the view")
    close(g_view.cpu().numpy()[:dw.size].reshape(dw.shape), dw, what="dW through the view")
    close(g_view.cpu().numpy()[dw.size:], db, what="db")
    L.cols_view = False
    dx_cols = L.backward(xt, yt, dyt, act=act)
    assert torch.equal(dx_cols, dx_view) and torch.equal(L.grad, g_view)
    # a batch that is not whole tiles is refused by the products themselves
    assert _lib.lib.rlgan_deconv_backward_input_view(L.spec, dyt.data_ptr(), L.wmat.data_ptr(), dx_view.data_ptr(), n - 1, None, 1, None) == -2


@pytest.mark.parametrize("n", [3, 64])
def test_decoder_layers_hand_dz_down(n):
    """The two fused layers return their input gradient times ReLU'(input) and the layer below skips its own ReLU' (at 64
    images the few-pixel layer below reads that dZ through the cols view): every gradient and the code gradient keep the
    bits of the pass in which every layer applies its own ReLU'."""
    import rl_gan_amd as R
    rng = np.random.RandomState(n)
    dec = R.Decoder(200, rng=rng, bias_init=0.01)
    code = torch.from_numpy(rng.standard_normal((n, 200)).astype(np.float32)).cuda()
    img = dec.forward(code)
    d = torch.from_numpy(rng.standard_normal(tuple(img.shape)).astype(np.float32)).cuda()
    assert any(L.hands_dz for L in dec.layers)
    dcode = dec.backward(d).clone()
    grads = [L.grad.clone() for L in dec.layers]
    for L in dec.layers:
        L.hands_dz = False
    dcode2 = dec.backward(d)
    assert torch.equal(dcode, dcode2)
    for L, g in zip(dec.layers, grads):
        assert torch.equal(L.grad, g)


def test_decoder_end_to_end():
    import rl_gan_amd as R
    from gpu_util import close
    from oracle import deconv as od
    rng = np.random.RandomState(1)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    params = [(f32(w), f32(b)) for w, b in od.init_decoder(rng, 200, bias=0.001)]
    code = f32(np.abs(rng.standard_normal((6, 200))))          # the decoder's input is a ReLU output (Conv_model.py:455)
    img, caches = od.decoder_forward(code, params)
    dimg = f32(rng.standard_normal(img.shape))
    dcode, grads = od.decoder_backward(dimg, params, caches)
    dec = R.Decoder(200, params=params)
    assert dec.out_shape == (42, 42, 1)
    out = dec.forward(torch.from_numpy(code.astype(np.float32)).cuda())
    close(out.cpu().numpy(), img, what="image")
    dc = dec.backward(torch.from_numpy(dimg.astype(np.float32)).cuda())
    close(dc.cpu().numpy(), dcode, what="dcode")
    for i, (L, (dw, db)) in enumerate(zip(dec.layers, grads)):
        g = L.grad.cpu().numpy()
        close(g[:dw.size].reshape(dw.shape), dw, what=f"layer {i} dW")
        close(g[dw.size:], db, what=f"layer {i} db")


def test_deconv_argument_errors():
    from rl_gan_amd import _lib
    g = _lib.DeconvGeom()
    assert _lib.lib.rlgan_deconv_geometry(_lib.Deconv(3, 3, 6, 8, 3, 2), C.byref(g)) == -2        # c_in not a multiple of 4
    assert _lib.lib.rlgan_deconv_geometry(_lib.Deconv(3, 3, 8, 3, 3, 2), C.byref(g)) == -2        # k*k*c_out not a multiple of 4
    assert _lib.lib.rlgan_deconv_geometry(_lib.Deconv(19, 19, 32, 1, 6, 2), C.byref(g)) == 0
    assert (g.h_out, g.w_out, g.kk, g.n_params) == (42, 42, 36, 36 * 32 + 1)
    d = _lib.Deconv(19, 19, 32, 1, 6, 2)
    x = torch.zeros(64, device="cuda")
    assert _lib.lib.rlgan_deconv_place(d, None, x.data_ptr(), x.data_ptr(), 1, 0, None) == -1
    assert _lib.lib.rlgan_deconv_place(d, x.data_ptr(), x.data_ptr(), x.data_ptr(), 70000, 0, None) == -1
    assert _lib.lib.rlgan_deconv_gather(d, x.data_ptr(), None, None, 1, None) == -1
    assert _lib.lib.rlgan_column_sums(x.data_ptr(), None, 4, 300, x.data_ptr(), x.data_ptr(), None) == -2
    assert _lib.lib.rlgan_gather_map(x.data_ptr(), None, 4, x.data_ptr(), None) == -1
    torch.cuda.synchronize()
