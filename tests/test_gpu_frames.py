"""uint8 frame replay on the device (rlgan_frames_store / rlgan_frames_gather, FrameMemory) against the
reference-produced fixture and the oracle: slots, indices and the scaled float32 frames bit for bit.
Reference: Conv_model.py:102-157, env_wrapper.py:44,58."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import frames as OF

pytestmark = pytest.mark.gpu


def test_fixture_replay_bit_exact():
    import rl_gan_amd as R
    d = np.load(os.path.join(GOLDEN, "pixel_memory.npz"))
    cap, shape = int(d["capacity"]), d["frames_s"].shape[1:]
    m = R.FrameMemory(cap, shape)
    for i, (s, s_) in enumerate(zip(d["frames_s"], d["frames_s_"])):
        m.store_transition(s, s_)
        assert m.pointer == d["pointers"][i] and m.len == d["lens"][i]
    all_slots = torch.arange(cap + 1, device="cuda")
    np.testing.assert_array_equal(m.gather(all_slots, "s").cpu().numpy(), d["data_s"])
    np.testing.assert_array_equal(m.gather(all_slots, "s_").cpu().numpy(), d["data_s_"])
    np.random.seed(int(d["sample_seed"]))
    idx, b_s, b_s_ = m.sample(len(d["indices"]))
    np.testing.assert_array_equal(idx.cpu().numpy(), d["indices"])
    np.testing.assert_array_equal(b_s.cpu().numpy(), d["b_s"])
    np.testing.assert_array_equal(b_s_.cpu().numpy(), d["b_s_"])
    # the global stream moved on exactly as in the reference
    np.random.seed(int(d["sample_seed"])); np.random.choice(cap + 1, size=len(d["indices"]))
    ref_next = np.random.randint(1 << 30)
    np.random.seed(int(d["sample_seed"])); m.sample(len(d["indices"]))
    assert np.random.randint(1 << 30) == ref_next


@pytest.mark.parametrize("chunk", [1, 5, 16])
def test_batched_stores_match_oracle(chunk):
    import rl_gan_amd as R
    rs = np.random.RandomState(4)
    cap, shape, n = 16, (12, 10, 4), 61
    fs = rs.randint(0, 256, (n,) + shape).astype(np.uint8)
    fs_ = rs.randint(0, 256, (n,) + shape).astype(np.uint8)
    o = OF.FrameMemory(cap)
    m = R.FrameMemory(cap, shape, rng=R.MT19937Device(9))
    for i in range(0, n, chunk):
        for j in range(i, min(i + chunk, n)):
            o.store_transition(fs[j], fs_[j])
        m.store_transitions(torch.from_numpy(fs[i:i + chunk]).cuda(), fs_[i:i + chunk])
        assert (m.pointer, m.len) == (o.pointer, o.len)
        sl = torch.arange(m.len, device="cuda")
        np.testing.assert_array_equal(m.gather(sl, "s").cpu().numpy(), np.array(o.data_s))
        np.testing.assert_array_equal(m.gather(sl, "s_").cpu().numpy(), np.array(o.data_s_))
    idx, b_s, b_s_ = m.sample(37)
    oi, ob, ob_ = o.sample(37, np.random.RandomState(9))
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_array_equal(b_s.cpu().numpy(), ob)
    np.testing.assert_array_equal(b_s_.cpu().numpy(), ob_)


def test_all_byte_values_scale_like_numpy():
    import rl_gan_amd as R
    m = R.FrameMemory(3, (16, 16, 1))
    f = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1)
    m.store_transitions(f, f[:, ::-1].copy())
    z = torch.zeros(1, dtype=torch.int64, device="cuda")
    np.testing.assert_array_equal(m.gather(z, "s").cpu().numpy(), OF.scaled(f))
    np.testing.assert_array_equal(m.gather(z, "s_").cpu().numpy(), OF.scaled(f[:, ::-1]))


def test_full_size_minibatch_properties():
    """84x84x4 stacks (doom/mario shape of the reference), minibatch 4096: every gathered frame equals the
    host's uint8 * (1/255) of the slot its index names, and out-of-range indices give zero frames."""
    import rl_gan_amd as R
    cap, shape, B = 2000, (84, 84, 4), 4096
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    m = R.FrameMemory(cap, shape, rng=R.MT19937Device(2))
    for i in range(0, cap + 1, 667):
        k = min(667, cap + 1 - i)
        f = torch.randint(0, 256, (k,) + shape, dtype=torch.uint8, device="cuda", generator=g)
        m.store_transitions(f, 255 - f)
    assert m.len == cap + 1
    idx, b_s, b_s_ = m.sample(B)
    np.testing.assert_array_equal(idx.cpu().numpy(), np.random.RandomState(2).choice(cap + 1, size=B))
    scale = torch.tensor(np.float32(OF.SCALE), device="cuda")
    want = m._s[idx].to(torch.float32) * scale
    assert torch.equal(b_s.reshape(B, -1), want)
    assert torch.equal(b_s_.reshape(B, -1), (255 - m._s[idx]).to(torch.float32) * scale)
    bad = torch.tensor([-1, cap + 1, 3], dtype=torch.int64, device="cuda")
    out = m.gather(bad)
    assert float(out[:2].abs().max()) == 0.0 and torch.equal(out[2].reshape(-1), m._s[3].to(torch.float32) * scale)
    # a store whose slot index is out of range writes nothing
    before = m._s.clone()
    from rl_gan_amd._lib import check, lib
    check(lib.rlgan_frames_store(m._s.data_ptr(), m.slots, m.frame_bytes, before.data_ptr(), 2,
                                     torch.tensor([-1, cap + 1], dtype=torch.int64, device="cuda").data_ptr(), None), "store")
    torch.cuda.synchronize()
    assert torch.equal(m._s, before)


def test_one_channel_and_both_fields_in_place():
    """`gather_channel` is the slice of `gather` (Curious_GAN_mario.py:92 takes the newest frame of s'), channels counted
    from either end, out-of-range slots as zeros; `ref_pair` names s and s' of the same slots as one batch whose encoder
    features equal those of the gathered float32 frames bit for bit."""
    import rl_gan_amd as R
    rs = np.random.RandomState(8)
    cap, shape, n = 20, (42, 42, 4), 20
    m = R.FrameMemory(cap, shape, rng=R.MT19937Device(2))
    m.store_transitions(rs.randint(0, 256, (n,) + shape).astype(np.uint8), rs.randint(0, 256, (n,) + shape).astype(np.uint8))
    idx = torch.from_numpy(rs.randint(0, n, 33)).cuda()
    full = {w: m.gather(idx, w) for w in ("s", "s_")}
    assert torch.equal(m.gather_channel(idx), full["s_"][..., 3:])
    assert torch.equal(m.gather_channel(idx, 0, "s"), full["s"][..., :1])
    assert torch.equal(m.gather_channel(idx, -3, "s"), full["s"][..., 1:2])
    bad = torch.tensor([3, -1, cap + 1, 5], dtype=torch.int64, device="cuda")
    g = m.gather_channel(bad, 2, "s")
    assert not g[1].any() and not g[2].any() and torch.equal(g[0], m.gather(bad[:1], "s")[0, ..., 2:3])
    odd = R.FrameMemory(6, (4, 4, 3))                         # three channels: a pixel is not a word
    f = rs.randint(0, 256, (6, 4, 4, 3)).astype(np.uint8)
    odd.store_transitions(f, f)
    six = torch.arange(6, device="cuda")
    assert torch.equal(odd.gather_channel(six, 1, "s"), odd.gather(six, "s")[..., 1:2])
    enc = R.ConvEncoder(42, 42, 4, R.universe_head(2), rng=rs, dense=256, bias_init=0.001)
    pair = m.ref_pair(idx)
    assert pair.n == 66 and torch.equal(pair.idx[:33], idx) and torch.equal(pair.idx[33:], idx + m.slots)
    want = enc.forward(torch.cat([full["s"], full["s_"]], 0)).clone()
    assert torch.equal(enc.forward(pair), want)
    buf = torch.zeros(66, dtype=torch.int64, device="cuda")
    again = m.ref_pair(idx, out=buf)
    assert again.idx.data_ptr() == buf.data_ptr() and torch.equal(buf, pair.idx)


def test_argument_errors():
    import rl_gan_amd as R
    from rl_gan_amd._lib import lib
    ring = torch.zeros(4 * 64 + 16, dtype=torch.uint8, device="cuda")
    out = torch.zeros(4 * 64 + 8, dtype=torch.float32, device="cuda")
    idx = torch.zeros(4, dtype=torch.int64, device="cuda")
    E_ARG, E_ALIGN = -1, -5
    p, o, i = ring.data_ptr(), out.data_ptr(), idx.data_ptr()
    assert lib.rlgan_frames_gather(None, 4, 64, i, 4, 1.0, o, None) == E_ARG
    assert lib.rlgan_frames_gather(p, 0, 64, i, 4, 1.0, o, None) == E_ARG
    assert lib.rlgan_frames_gather(p, 4, 64, i, 0, 1.0, o, None) == E_ARG
    assert lib.rlgan_frames_gather(p, 4, 64, i, 65536, 1.0, o, None) == E_ARG
    assert lib.rlgan_frames_gather(p, 4, 62, i, 4, 1.0, o, None) == E_ALIGN
    assert lib.rlgan_frames_gather(p + 1, 4, 64, i, 4, 1.0, o, None) == E_ALIGN
    assert lib.rlgan_frames_gather(p, 4, 64, i, 4, 1.0, o + 4, None) == E_ALIGN
    assert lib.rlgan_frames_gather_channel(p, 4, 16, 4, 4, i, 4, 1.0, o, None) == -2          # channel outside the frame's
    assert lib.rlgan_frames_gather_channel(p, 4, 16, 0, 0, i, 4, 1.0, o, None) == -2
    assert lib.rlgan_frames_gather_channel(p, 4, 16, 4, 3, None, 4, 1.0, o, None) == E_ARG
    assert lib.rlgan_frames_gather_channel(p, 4, 16, 4, 3, i, 4, 1.0, o, None) == 0
    assert lib.rlgan_frames_store(p, 4, 64, None, 1, i, None) == E_ARG
    assert lib.rlgan_frames_store(p, 4, 63, p, 1, i, None) == E_ALIGN
    assert lib.rlgan_frames_store(p, 4, 64, p + 2, 1, i, None) == E_ALIGN
    torch.cuda.synchronize()


@pytest.mark.parametrize("spec", ["doom", "universe"])
def test_encoder_reads_uint8_ring_in_place(spec):
    """First conv layer on the uint8 ring through the index array == the same layer on the gathered float32
    frames, bit for bit (forward, ELU output, weight gradient), Conv_model.py:15-24,36-56."""
    import rl_gan_amd as R
    cap, hw, n = 50, 42, 37
    layers = R.DOOM_HEAD[:3] if spec == "doom" else R.universe_head(3)
    enc = R.ConvEncoder(hw, hw, 4, layers, rng=np.random.RandomState(3), bias_init=0.001, dense=64)
    mem = R.FrameMemory(cap, (hw, hw, 4), rng=R.MT19937Device(6))
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    f = torch.randint(0, 256, (cap,) + (hw, hw, 4), dtype=torch.uint8, device="cuda", generator=g)
    mem.store_transitions(f, f.flip(0))
    idx = mem.sample_indices(n)
    for which in ("s", "s_"):
        x = mem.gather(idx, which)
        feat_a = enc.forward(x).clone()
        d = torch.randn(feat_a.shape, device="cuda", generator=g)
        enc.backward(d)
        grads_a = [L.grad.clone() for L in enc.layers] + [enc.head.grad.clone()]
        feat_b = enc.forward(mem.ref(idx, which))
        assert torch.equal(feat_a, feat_b)
        assert enc.backward(d) is None
        for ga, L in zip(grads_a, enc.layers + [enc.head]):
            assert torch.equal(ga, L.grad)
    # no index array: image i is slot i; an index out of range reads as a zero image
    L0 = enc.layers[0]
    y_all = L0.forward_frames(R.FrameRef(mem._s, None, mem.SCALE, n=5))
    assert torch.equal(y_all, L0.forward(mem.gather(torch.arange(5, device="cuda"))))
    bad = torch.tensor([cap + 1, 2, -3], dtype=torch.int64, device="cuda")
    y_bad = L0.forward_frames(mem.ref(bad))
    assert torch.equal(y_bad, L0.forward(mem.gather(bad)))


def test_u8_conv_argument_errors():
    import rl_gan_amd as R
    from rl_gan_amd._lib import lib
    L = R.ConvLayer(12, 12, 4, 8, 3, 2, rng=np.random.RandomState(0))
    ring = torch.zeros((4, 12 * 12 * 4 + 4), dtype=torch.uint8, device="cuda")
    y = torch.zeros(8 * L.h_out * L.w_out * 8, dtype=torch.float32, device="cuda")
    E_ARG, E_ALIGN = -1, -5
    r, w, yo = ring.data_ptr(), L.wfrag.data_ptr(), y.data_ptr()
    assert lib.rlgan_conv_forward_u8(L.conv, None, 4, None, 1.0, w, yo, 2, 1, None) == E_ARG
    assert lib.rlgan_conv_forward_u8(L.conv, r, 0, None, 1.0, w, yo, 2, 1, None) == E_ARG
    assert lib.rlgan_conv_forward_u8(L.conv, r, 4, None, 1.0, w, yo, 5, 1, None) == E_ARG      # more images than slots
    assert lib.rlgan_conv_forward_u8(L.conv, r + 2, 4, None, 1.0, w, yo, 2, 1, None) == E_ALIGN
    assert lib.rlgan_conv_backward_weights_u8(L.conv, r, 4, None, 1.0, None, None, yo, 4, L.grad.data_ptr(), 2, None) == E_ARG
    torch.cuda.synchronize()


def test_pixel_memory_returns_what_the_reference_memory_returned():
    """All nine fields through the device replay against the arrays the reference's own `Conv_model.Memory.sample`
    produced for the same stores and NumPy seed (tests/golden/pixel_memory_full.npz): bit for bit, float frames
    accepted as uint8 * 1/255, the index draw taken from NumPy's global stream like the reference's."""
    import os
    import rl_gan_amd as R
    F = np.load(os.path.join(os.path.dirname(__file__), "golden", "pixel_memory_full.npz"))
    M = R.PixelMemory(int(F["capacity"]), shape=F["frames_s"].shape[1:], a_dim=int(F["ac"]), lstm_units=int(F["units"]), dense_dim=1)
    for i in range(F["frames_s"].shape[0]):
        s = F["frames_s"][i].astype(np.float32) * (1.0 / 255)              # env_wrapper.py:58
        s_ = F["frames_s_"][i] if i % 2 else F["frames_s_"][i].astype(np.float32) * (1.0 / 255)
        M.store_transition(s, F["g"][i], [F["lstm_s"][i, 0], F["lstm_s"][i, 1]], F["a"][i], float(F["r"][i]), F["curious_r"][i], s_,
                           [F["lstm_s_"][i, 0], F["lstm_s_"][i, 1]], int(F["dense_a"][i]))
    assert M.pointer == int(F["pointer"]) and M.len == int(F["len"])
    np.random.seed(int(F["sample_seed"]))
    out = M.sample(10)
    after = np.random.random()
    np.random.seed(int(F["sample_seed"])); np.random.choice(M.len, size=10)
    assert after == np.random.random()                                     # the global stream moved exactly like the reference's
    names = ("b_s", "b_g", "b_lstm_s", "b_a", "b_r", "b_curious_r", "b_s_", "b_lstm_s_again", "b_d_a")
    assert len(out) == 9
    for k, v in zip(names, out):
        assert np.asarray(v).shape == F[k].shape, k
        np.testing.assert_array_equal(np.asarray(v, dtype=np.float64), F[k].astype(np.float64), err_msg=k)
    # the training loop's dense action is a vector as long as the action (Curious_GAN_mario.py:267-270)
    M2 = R.PixelMemory(3, shape=F["frames_s"].shape[1:], a_dim=int(F["ac"]), lstm_units=int(F["units"]))
    dense = np.ones(int(F["ac"])) * 0.1; dense[1] = 1
    M2.store_transition(F["frames_s"][0], F["g"][0], [F["lstm_s"][0, 0], F["lstm_s"][0, 1]], F["a"][0], 1.0, F["curious_r"][0], F["frames_s_"][0],
                        [F["lstm_s_"][0, 0], F["lstm_s_"][0, 1]], dense)
    np.testing.assert_array_equal(M2.sample(2)[8], np.tile(dense.astype(np.float32), (2, 1)))
    d = M.sample_device(4)
    assert d["s"].shape == (4,) + F["frames_s"].shape[1:] and d["curious_r"].shape == (4, 1) and d["a"].shape == (4, int(F["ac"]))
    with pytest.raises(ValueError):
        M.store_transition(np.full(F["frames_s"].shape[1:], 0.3, np.float32), F["g"][0], [F["lstm_s"][0, 0], F["lstm_s"][0, 1]], F["a"][0],
                           0.0, F["curious_r"][0], F["frames_s_"][0], [F["lstm_s_"][0, 0], F["lstm_s_"][0, 1]], 0)
