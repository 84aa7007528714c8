"""Device forward passes (choose_action / predict / determine / q) against the float64 oracle.
Reference: DDPG.py:78-95,103-105,153-172; Curious_GAN.py:46-82,87-92,112-135,150-153."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [1, 16, 37, 4096])
def test_forward_all_networks(B):
    from gpu_util import Pair, close, dev
    from oracle import nets
    from rl_gan_amd.core import mlp_forward
    p = Pair()
    rng = np.random.RandomState(B)
    s = rng.uniform(-1.5, 1.5, (B, 7)).astype(np.float32).astype(np.float64)
    a = rng.uniform(-1, 1, (B, 2)).astype(np.float32).astype(np.float64)
    x = rng.uniform(-1.5, 1.5, (B, 7)).astype(np.float32).astype(np.float64)
    close(mlp_forward(p.A.e, dev(s)).cpu().numpy(), p.o.choose_action(s), what="actor eval")
    at, _ = nets.forward(p.o.actor_spec, p.o.actor_t, {"s": s})
    close(mlp_forward(p.A.t, dev(s)).cpu().numpy(), at, what="actor target")
    q, _ = nets.forward(p.o.critic_spec, p.o.critic_e, {"s": s, "a": a})
    close(mlp_forward(p.C.e, dev(s), a=dev(a)).cpu().numpy(), q, what="critic")
    close(mlp_forward(p.G.net, dev(s), a=dev(a)).cpu().numpy(), p.o.predict(s, a), what="generator")
    close(mlp_forward(p.D.net, dev(s), a=dev(a), x=dev(x), out_mode=2).cpu().numpy(), p.o.determine(s, a, x), what="determine")
    close(mlp_forward(p.D.net, dev(s), a=dev(a), x=dev(x), out_mode=1).cpu().numpy(), p.o.d_prob(s, a, x), what="D prob")


def test_forward_strided_inputs_and_weight_roundtrip():
    from gpu_util import Pair, close, dev, make_batch
    from rl_gan_amd.core import mlp_forward
    p = Pair(spread=False)
    b = make_batch(64, 3)
    bt = dev(b)                                     # [64,25], fields are strided views
    close(mlp_forward(p.G.net, bt[:, :7], a=bt[:, 21:23]).cpu().numpy(), p.o.predict(b[:, :7], b[:, 21:23]), what="G strided")
    flat = p.G.net.get_flat()
    np.testing.assert_array_equal(flat, p.o.flat("gen").astype(np.float32))
    t = p.G.net.get_tensors()
    assert list(t.keys())[0] == "l1/w1_s" and t["l3/kernel"].shape == (200, 50)
    p.G.net.set_tensors(t)
    np.testing.assert_array_equal(p.G.net.get_flat(), flat)


def test_bad_arguments_are_reported_not_crashed():
    from gpu_util import Pair, dev
    from rl_gan_amd import _lib
    from rl_gan_amd.core import mat, _stream
    p = Pair(spread=False)
    out = torch.zeros((4, 2), device="cuda")
    rc = _lib.lib.rlgan_mlp_forward(9, 7, 2, p.A.e.net(), mat(dev(np.zeros((4, 7)))), mat(None), mat(None), out.data_ptr(), 2, 0, 4, _stream())
    assert rc == -3
    rc = _lib.lib.rlgan_mlp_forward(0, 40, 2, p.A.e.net(), mat(dev(np.zeros((4, 7)))), mat(None), mat(None), out.data_ptr(), 2, 0, 4, _stream())
    assert rc == -2
    with pytest.raises(_lib.RlganError):
        _lib.check(_lib.lib.rlgan_mlp_forward(1, 7, 2, p.C.e.net(), mat(dev(np.zeros((4, 7)))), mat(None), mat(None), out.data_ptr(), 2, 0, 4, _stream()))
