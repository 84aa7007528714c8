"""The pixel generator's training step on the device (Conv_model.py:437-475,522-541; Curious_GAN_mario.py:340-353)
against the float64 restatement: predicted image, loss, every gradient of the generator's variables
(1e-5 relative to the tensor scale, fp32), and the Adam update given those gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n, a_dim, seed=0):
    import rl_gan_amd as R
    from oracle import conv as oc, deconv as od
    rng = np.random.RandomState(seed)
    f32 = lambda v: np.asarray(v).astype(np.float32).astype(np.float64)
    h_dim = 256
    enc_params = [(f32(rng.uniform(-0.3, 0.3, (3, 3, 1, 32))), f32(rng.uniform(-0.1, 0.1, 32))),
                  (f32(rng.uniform(-0.1, 0.1, (3, 3, 32, 32))), f32(rng.uniform(-0.1, 0.1, 32)))]
    feat = 9 * 9 * 32                                       # 42 -> 20 -> 9
    enc_dense = (f32(rng.uniform(-0.03, 0.03, (feat, 256))), f32(np.full(256, 0.001)))
    w1_s, w1_a = f32(rng.uniform(-0.1, 0.1, (h_dim, 200))), f32(rng.uniform(-0.2, 0.2, (a_dim, 200)))
    b1 = f32(np.full(200, 0.001))
    dec_params = [(f32(w), f32(b)) for w, b in od.init_decoder(rng, 200, bias=0.0)]
    phis = f32(rng.standard_normal((n, h_dim)))
    a = f32(rng.uniform(-1, 1, (n, a_dim)))
    s_next = f32(rng.uniform(0, 1, (n, 42, 42, 1)))
    enc = R.single_frame_encoder(42, 2, params=enc_params, dense_params=enc_dense)
    gen = R.PixelGenerator(h_dim, a_dim, 1e-4, enc, w1_s=w1_s, w1_a=w1_a, b1=b1, decoder_params=dec_params)
    host = dict(phis=phis, a=a, s_next=s_next, w1_s=w1_s, w1_a=w1_a, b1=b1, dec_params=dec_params, enc_params=enc_params,
                enc_dense=enc_dense)
    dev = [torch.from_numpy(v.astype(np.float32)).cuda() for v in (phis, a, s_next)]
    return gen, enc, host, dev


@pytest.mark.parametrize("n,a_dim", [(6, 6), (3, 14)])
def test_generator_step_matches_restatement(n, a_dim):
    from gpu_util import close
    from oracle import pixel_gan as op
    from oracle.optim import Adam
    gen, enc, H, (phis, a, s_next) = _setup(n, a_dim)
    loss, image, g = op.generator_loss_and_grads(H["phis"], H["a"], H["s_next"], H["w1_s"], H["w1_a"], H["b1"], H["dec_params"],
                                                 H["enc_params"], H["enc_dense"])
    enc_before = [L.canon.clone() for L in enc.layers] + [enc.head.canon.clone()]
    layers = [gen.l1] + gen.decoder.layers
    before = [L.canon.cpu().numpy().copy() for L in layers]
    assert abs(gen.eval(phis, a, s_next) - loss) <= 1e-5 * abs(loss)
    out = gen.learn(phis, a, s_next)
    close(gen._img.cpu().numpy(), image, what="image")
    assert abs(float(out.item()) - loss) <= 1e-5 * abs(loss)
    g1 = gen.l1.grad.cpu().numpy()
    k = gen.k_pad
    W = g1[:k * 200].reshape(k, 200)
    close(W[:256], g["w1_s"], what="dw1_s")
    close(W[256:256 + a_dim], g["w1_a"], what="dw1_a")
    assert not W[256 + a_dim:].any()
    close(g1[k * 200:], g["b1"], what="db1")
    for i, (L, (dw, db)) in enumerate(zip(gen.decoder.layers, g["decoder"])):
        gl = L.grad.cpu().numpy()
        close(gl[:dw.size].reshape(dw.shape), dw, what=f"decoder {i} dW")
        close(gl[dw.size:], db, what=f"decoder {i} db")
    # Adam (Conv_model.py:531) on the device's own gradients; the images the kernels read follow canon
    for L, w0 in zip(layers, before):
        opt = Adam(w0.size, 1e-4, dtype=np.float32)
        want = opt.step(w0, L.grad.cpu().numpy())
        np.testing.assert_allclose(L.canon.cpu().numpy(), want, rtol=0, atol=2e-7)
    for L in gen.decoder.layers:
        kern, _ = L.kernel_bias()
        if L.fused:                                         # the two fragment images of the fused kernels
            nk = L.kk * L.c_in
            for m, img in ((L.map_f, L.wfrag), (L.map_dx, L.wdx)):
                np.testing.assert_array_equal(img.cpu().numpy()[m.cpu().numpy()[:nk]], kern.ravel())
        else:
            wm = L.wmat.cpu().numpy()[:L.c_in * L.kk].reshape(L.c_in, L.kk)
            np.testing.assert_array_equal(wm, kern.reshape(L.kk, L.c_in).T)
    assert [L.fused for L in gen.decoder.layers] == [False, True, True, True] and [L.fused_bwd for L in gen.decoder.layers] == [False, False, True, True]
    # the encoder is not in the generator's var_list
    for t, L in zip(enc_before, enc.layers + [enc.head]):
        assert torch.equal(t, L.canon)
    assert int(gen.step.item()) == 1


def test_generator_loss_falls():
    gen, enc, H, (phis, a, s_next) = _setup(8, 6, seed=3)
    gen.lr = 1e-3
    l0 = gen.eval(phis, a, s_next)
    for _ in range(30):
        gen.learn(phis, a, s_next)
    assert gen.eval(phis, a, s_next) < 0.7 * l0 and int(gen.step.item()) == 30


def test_glue_argument_errors():
    from rl_gan_amd._lib import lib
    x = torch.zeros(64, device="cuda")
    p = x.data_ptr()
    assert lib.rlgan_copy_cols(p, 1, None, 4, 4, 1, 3, None) == -1
    assert lib.rlgan_copy_cols(p, 1, p, 2, 4, 1, 3, None) == -5          # destination rows too short
    assert lib.rlgan_mse(p, None, 8, p, None, None) == -1
    assert lib.rlgan_relu_backward(p, p, p, 0, None) == -1
    torch.cuda.synchronize()
