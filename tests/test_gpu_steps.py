"""The four update steps and the post-update evaluations on the device against the float64 oracle:
per-step losses, every gradient tensor (flat, TF variable order) and the updated weights.
Reference: DDPG.py:97-113,141-151,174-178; Curious_GAN.py:33-45,84-95,137-148,199-213."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fields(b):
    return dict(s=b[:, :7], s_=b[:, 7:14], g=b[:, 14:21], a=b[:, 21:23], r=b[:, 23:24], cr=b[:, 24:25])


@pytest.mark.parametrize("B", [16, 32, 100, 4096])
def test_each_step_gradients_losses_and_weights(B):
    from gpu_util import Pair, close, dev, make_batch
    p = Pair()
    b = make_batch(B, seed=B)
    f = _fields(b)
    bt = dev(b)
    ft = dict(s=bt[:, :7], s_=bt[:, 7:14], g=bt[:, 14:21], a=bt[:, 21:23], r=bt[:, 23], cr=bt[:, 24])
    eng, o = p.eng, p.o
    w0 = p.snapshot()

    # (1) critic step, including the continuation into the actor eval net
    ref = o.critic_learn(f["s"], f["a"], f["cr"], f["s_"])
    wc, wa = eng.critic_step(p.C, p.A, ft["s"], ft["a"], ft["cr"], ft["s_"])
    close(wc.grad.cpu().numpy(), ref["g_critic"], what="critic grads")
    close(wa.grad.cpu().numpy(), ref["g_actor"], what="actor grads through the critic loss")
    close(eng.losses[eng.LOSS_CRITIC].item(), ref["loss"], what="TD loss")
    p.check_weights("after critic step")
    p.check_deltas(w0, "critic step")
    assert int(p.C.counter.item()) == 1

    # (2) actor step
    ref = o.actor_learn(f["s"])
    wa2 = eng.actor_step(p.A, p.C, ft["s"])
    close(wa2.grad.cpu().numpy(), ref["g_actor"], what="policy grads")
    p.check_weights("after actor step")
    p.check_deltas(w0, "critic + actor step")

    # (3) discriminator step
    ref = o.disc_learn(f["g"], f["s_"], f["s"], f["a"])
    wd = eng.disc_step(p.D, ft["g"], ft["s_"], ft["s"], ft["a"])
    close(wd.grad.cpu().numpy(), ref["g_disc"], what="D grads")
    close(eng.losses[eng.LOSS_D_STEP].item(), ref["loss"], what="D loss (pre-update)")
    p.check_weights("after D step")

    # (4) generator step
    ref = o.gen_learn(f["s"], f["a"])
    wg = eng.gen_step(p.G, p.D, ft["s"], ft["a"])
    close(wg.grad.cpu().numpy(), ref["g_gen"], what="G grads")
    close(eng.losses[eng.LOSS_G_STEP].item(), ref["loss"], what="G loss (pre-update)")
    p.check_weights("after G step")
    p.check_deltas(w0, "all four steps")
    p.check_slots("after the four steps")

    # (5), (6) post-update losses
    out = eng.gan_eval(p.G, p.D, ft["g"], ft["s_"], ft["s"], ft["a"]).cpu().numpy()
    close(out[0], o.disc_eval(f["g"], f["s_"], f["s"], f["a"]), what="D_loss eval")
    close(out[1], o.gen_eval(f["s"], f["a"]), what="G_loss eval")


def test_each_step_at_batch_32768():
    """The global batch of the eight-GPU configuration on one GPU: every step from the SAME initial weights (four agents),
    on a minibatch whose rows sit clear of the activations' corners for those weights.  Rows at which a backward pass meets
    a pre-activation within float32 rounding of a corner of relu / relu6 / leaky relu are drawn again (about 1.4 % of
    them): there the derivative mask hangs on the last bit of a float32 sum, a float64 oracle cannot say which side is
    right, and in 32,768 rows some always are (oracle_util.kink_rows).  Steps chained through updated weights, as in
    the test above, cannot be cleared this way: the weights a later step sees depend on the rows themselves."""
    from gpu_util import Pair, close, dev, make_smooth_batch
    B = 32768
    p = Pair()
    b, n_redrawn = make_smooth_batch(p.o, B, seed=B)
    assert 0 < n_redrawn < B // 20
    f, bt = _fields(b), dev(b)
    ft = dict(s=bt[:, :7], s_=bt[:, 7:14], g=bt[:, 14:21], a=bt[:, 21:23], cr=bt[:, 24])
    ref = p.o.critic_learn(f["s"], f["a"], f["cr"], f["s_"])
    wc, wa = p.eng.critic_step(p.C, p.A, ft["s"], ft["a"], ft["cr"], ft["s_"])
    close(wc.grad.cpu().numpy(), ref["g_critic"], what="critic grads")
    close(wa.grad.cpu().numpy(), ref["g_actor"], what="actor grads through the critic loss")
    close(p.eng.losses[p.eng.LOSS_CRITIC].item(), ref["loss"], what="TD loss")
    p.check_weights("after critic step")
    p = Pair()
    ref = p.o.actor_learn(f["s"])
    close(p.eng.actor_step(p.A, p.C, ft["s"]).grad.cpu().numpy(), ref["g_actor"], what="policy grads")
    p.check_weights("after actor step")
    p = Pair()
    ref = p.o.disc_learn(f["g"], f["s_"], f["s"], f["a"])
    close(p.eng.disc_step(p.D, ft["g"], ft["s_"], ft["s"], ft["a"]).grad.cpu().numpy(), ref["g_disc"], what="D grads")
    close(p.eng.losses[p.eng.LOSS_D_STEP].item(), ref["loss"], what="D loss (pre-update)")
    p.check_weights("after D step")
    p = Pair()
    ref = p.o.gen_learn(f["s"], f["a"])
    close(p.eng.gen_step(p.G, p.D, ft["s"], ft["a"]).grad.cpu().numpy(), ref["g_gen"], what="G grads")
    close(p.eng.losses[p.eng.LOSS_G_STEP].item(), ref["loss"], what="G loss (pre-update)")
    p.check_weights("after G step")
    out = p.eng.gan_eval(p.G, p.D, ft["g"], ft["s_"], ft["s"], ft["a"]).cpu().numpy()
    close(out[0], p.o.disc_eval(f["g"], f["s_"], f["s"], f["a"]), what="D_loss eval")
    close(out[1], p.o.gen_eval(f["s"], f["a"]), what="G_loss eval")


@pytest.mark.parametrize("B", [32, 4096])
def test_three_fused_updates_track_the_oracle(B):
    """Three consecutive fused updates (two-phase order) == the reference's sequential order; covers
    the first-call target sync and the second Adam/RMSProp step."""
    from gpu_util import Pair, close, dev, make_batch
    from oracle.agent import Hyper
    p = Pair(hyper=Hyper(replace_iter_a=2, replace_iter_c=3))
    for it in range(3):
        b = make_batch(B, seed=100 + it)
        ref = p.o.update(b)
        p.eng.update(p.A, p.C, p.G, p.D, dev(b))
        l = p.eng.losses.cpu().numpy()
        close(l[p.eng.LOSS_CRITIC], ref["critic_loss"], what=f"it{it} critic loss")
        close(l[p.eng.LOSS_D_EVAL], ref["D_loss"], what=f"it{it} D_loss")
        close(l[p.eng.LOSS_G_EVAL], ref["G_loss"], what=f"it{it} G_loss")
        p.check_weights(f"after update {it}")
        # strict after the first update; from the second on the generator's first moment is off by up to 1.2e-5 of its
        # scale at batch 4096 (2 of 52,607 entries, measured), the echo of noise-decided Adam steps in the weights
        p.check_slots(f"after update {it}", rtol=1e-5 if it == 0 else 3e-5)
    assert int(p.C.counter.item()) == 3 and int(p.A.counter.item()) == 3 and int(p.D.step.item()) == 3
    # target nets were copied on calls 0 (both) and 2 (actor) / 3 would be next for the critic
    np.testing.assert_array_equal(p.A.t.get_flat(), p.A.e.get_flat())
    assert not np.array_equal(p.C.t.get_flat(), p.C.e.get_flat())


def test_critic_step_without_the_actor_continuation():
    from gpu_util import Pair, close, dev, make_batch
    from oracle.agent import Hyper
    p = Pair(hyper=Hyper(critic_trains_actor=False))
    b = make_batch(48, 5)
    bt = dev(b)
    before = p.A.e.get_flat()
    ref = p.o.critic_learn(b[:, :7], b[:, 21:23], b[:, 24:25], b[:, 7:14])
    wc, wa = p.eng.critic_step(p.C, p.A, bt[:, :7], bt[:, 21:23], bt[:, 24], bt[:, 7:14])
    assert wa is None and "g_actor" not in ref
    close(wc.grad.cpu().numpy(), ref["g_critic"], what="critic grads")
    np.testing.assert_array_equal(p.A.e.get_flat(), before)
    p.check_weights("critic only")


def test_ddpg_only_update_uses_extrinsic_reward():
    from gpu_util import Pair, close, dev, make_batch
    p = Pair()
    b = make_batch(32, 9)
    ref = p.o.update_ddpg(b)
    p.eng.update_ddpg(p.A, p.C, dev(b))
    close(p.eng.losses[p.eng.LOSS_CRITIC].item(), ref["critic_loss"], what="DDPG critic loss")
    p.check_weights("after DDPG update")


def test_soft_target_update_and_tau_one_equals_hard_copy():
    from gpu_util import Pair
    from rl_gan_amd.core import target_update, mlp_forward
    p = Pair(spread=False)
    e, t = p.C.e.get_flat(), p.C.t.get_flat()
    target_update(p.C.t, p.C.e, 0.25, None, 1)
    np.testing.assert_allclose(p.C.t.get_flat(), 0.25 * e + 0.75 * t, rtol=1e-6, atol=1e-7)
    target_update(p.C.t, p.C.e, 1.0, None, 1)
    np.testing.assert_array_equal(p.C.t.get_flat(), e)
    s = torch.rand((16, 7), device="cuda", dtype=torch.float32); a = torch.rand((16, 2), device="cuda", dtype=torch.float32)
    # the fragment image followed the canonical one
    assert torch.equal(mlp_forward(p.C.t, s, a=a), mlp_forward(p.C.e, s, a=a))
