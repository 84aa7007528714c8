"""Launch-level fusions must not change results: minibatch rows read through an index from the ring,
the one-launch form of {losses, acting step, index draw}, and the in-kernel noise / goal draws are
compared bit for bit with the separate calls they replace (which the other tests hold against the
oracle).  Reference: Curious_GAN.py:168-169,190-213."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_update_through_ring_indices_equals_update_on_gathered_batch():
    import rl_gan_amd as R
    from gpu_util import Pair, dev, make_batch
    cap, B = 512, 100
    ring_rows = make_batch(cap, seed=3)
    mem = R.DeviceMemory(cap, 25, rng=R.MT19937Device(5))
    mem.store_rows(dev(ring_rows))
    idx = mem.sample_indices(B)
    batch = mem.gather(idx)
    a, b = Pair(seed=11), Pair(seed=11)
    for _ in range(2):
        a.eng.update(a.A, a.C, a.G, a.D, batch)
        b.eng.update(b.A, b.C, b.G, b.D, mem._data, idx=idx)
    assert torch.equal(a.eng.losses, b.eng.losses)
    for x, y in ((a.A.e, b.A.e), (a.A.t, b.A.t), (a.C.e, b.C.e), (a.G.net, b.G.net), (a.D.net, b.D.net)):
        np.testing.assert_array_equal(x.get_flat(), y.get_flat())


def _rollout_fixture(n, cap, seed):
    import rl_gan_amd as R
    from gpu_util import Pair
    p = Pair(seed=seed, spread=False)
    env = R.VecArmEnv(n, "hard")
    u = torch.from_numpy(np.random.RandomState(seed).rand(n, 2)).cuda()
    env.reset(u)
    mem = R.DeviceMemory(cap, 25, rng=R.MT19937Device(seed))
    return p, env, mem


def test_in_kernel_draws_and_episode_cut_equal_the_separate_kernels():
    """(a) noise and reset uniforms produced by rlgan_philox_* and fed in, episode cut by
    rlgan_env_autoreset; (b) everything inside rlgan_act_rows, counters advanced in place by the last
    workgroup; (c) the same with counters and noise scale handed over between two copies (no ticket).
    Same ring, states and counters."""
    from rl_gan_amd import _lib
    from rl_gan_amd.core import _stream
    lib, check = _lib.lib, _lib.check
    n, cap, seed, steps, max_ep = 100, 256, 77, 7, 3
    out = []
    for fused in (False, True, "handover"):
        p, env, mem = _rollout_fixture(n, cap, 4)
        var2 = torch.full((2, 1), 2.0, dtype=torch.float64, device="cuda")
        ctr2 = torch.zeros((2, 2), dtype=torch.int64, device="cuda")
        var, ctr = var2[0], ctr2[0]
        ep_t = torch.zeros(n, dtype=torch.int32, device="cuda")
        ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
        nres = torch.zeros(1, dtype=torch.int32, device="cuda")
        z = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
        u = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
        for t in range(steps):
            common = (7, 2, n, p.A.e.net(), p.G.net.net(), p.D.net.net(), env.handle(), env.state32.data_ptr())
            ring = (mem._data.data_ptr(), cap, 25, mem.ld)
            if fused == "handover":
                cur = t & 1
                act = _lib.Act(*common, None, var2[cur].data_ptr(), 0.01, *ring, 0, 0, ctr2[cur].data_ptr(), env.reward.data_ptr(),
                               env.done.data_ptr(), seed, ep_t.data_ptr(), max_ep, 1, None, nres.data_ptr(), 0.9, 0.1,
                               ctr2[cur ^ 1].data_ptr(), var2[cur ^ 1].data_ptr())
                check(lib.rlgan_act_rows(act, None, _stream()))
                var, ctr = var2[cur ^ 1], ctr2[cur ^ 1]
            elif fused:
                act = _lib.Act(*common, None, var.data_ptr(), 0.01, *ring, 0, 0, ctr.data_ptr(), env.reward.data_ptr(),
                               env.done.data_ptr(), seed, ep_t.data_ptr(), max_ep, 1, None, nres.data_ptr(), 0.9, 0.1)
                check(lib.rlgan_act_rows(act, ticket.data_ptr(), _stream()))
            else:
                check(lib.rlgan_philox_normal(z.data_ptr(), n, seed, 0, 0, ctr[1:].data_ptr(), n, _stream()))
                check(lib.rlgan_philox_uniform(u.data_ptr(), n, seed, 1, n, ctr[1:].data_ptr(), n, _stream()))
                act = _lib.Act(*common, z.data_ptr(), var.data_ptr(), 0.01, *ring, t * n, t, None, env.reward.data_ptr(),
                               env.done.data_ptr(), 0, None, 0, 1, None, None, 0.0, 0.0)
                check(lib.rlgan_act_rows(act, None, _stream()))
                check(lib.rlgan_env_autoreset(env.handle(), n, 1, env.done.data_ptr(), ep_t.data_ptr(), max_ep, u.data_ptr(),
                                              env.state32.data_ptr(), nres.data_ptr(), ctr.data_ptr(), _stream()))
                if (t + 1) * n > cap:
                    var.mul_(0.9).clamp_(min=0.1)
        torch.cuda.synchronize()
        assert int(ticket.item()) == 0
        out.append(dict(ring=mem._data.cpu().numpy(), state=env.state32.cpu().numpy(), goal=env.goal.cpu().numpy(),
                        ep_t=ep_t.cpu().numpy(), ctr=ctr.tolist(), nres=int(nres.item()), var=float(var.item())))
    a = out[0]
    for b in out[1:]:
        assert a["ctr"] == b["ctr"] == [steps * n, steps]
        assert a["nres"] == b["nres"] and a["nres"] >= 2 * n          # max_ep = 3: everyone was cut at least twice
        for k in ("ring", "state", "goal", "ep_t"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        assert abs(a["var"] - b["var"]) < 1e-15 and abs(b["var"] - 2.0 * 0.9 ** 5) < 1e-12   # steps 2..6 end beyond the ring size
    assert out[1]["var"] == out[2]["var"]


def test_one_launch_of_losses_acting_and_draw_equals_three_calls():
    import rl_gan_amd as R
    from gpu_util import dev, make_batch
    from rl_gan_amd import _lib
    from rl_gan_amd.core import _stream
    lib, check = _lib.lib, _lib.check
    n, cap, B = 64, 512, 64          # equal tile counts: the round-robin deal is exercised
    res = []
    for fused in (False, True):
        p, env, mem = _rollout_fixture(n, cap, 9)
        mem.store_rows(dev(make_batch(cap, seed=1)))
        idx = mem.sample_indices(B)
        nxt = torch.zeros(B, dtype=torch.int64, device="cuda")
        var = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        ticket = p.eng.ticket()
        d = mem._data
        p.eng.update(p.A, p.C, p.G, p.D, d, idx=idx, with_eval=False)
        # the losses are evaluated from the input tiles the update left behind: the acting step below
        # overwrites ring rows the minibatch may hold
        ev = p.eng.eval_args_after_update(p.G, p.D, B)
        if not fused:
            ref = p.eng.gan_eval(p.G, p.D, d[:, 14:21], d[:, 7:14], d[:, :7], d[:, 21:23], 0.0, idx).clone()
        z = torch.from_numpy(np.random.RandomState(2).randn(n, 2).astype(np.float32)).cuda()
        act = _lib.Act(7, 2, n, p.A.e.net(), p.G.net.net(), p.D.net.net(), env.handle(), env.state32.data_ptr(), z.data_ptr(),
                       var.data_ptr(), 0.01, d.data_ptr(), cap, 25, mem.ld, mem.pointer, 0, None, env.reward.data_ptr(),
                       env.done.data_ptr(), 0, None, 0, 1, None, None, 0.0, 0.0)
        smp = _lib.Sample(mem.rng.state.data_ptr(), cap, B, nxt.data_ptr())
        if fused:
            check(lib.rlgan_rollout_eval_rows(act, ev, smp, ticket.data_ptr(), _stream()))
        else:
            check(lib.rlgan_gan_eval_rows(ev, ticket.data_ptr(), _stream()))
            assert torch.equal(ref, p.eng.losses[3:5])          # same numbers as reading the ring through idx
            check(lib.rlgan_act_rows(act, None, _stream()))
            mem.rng.choice(cap, B, nxt)
        torch.cuda.synchronize()
        res.append((p.eng.losses.cpu().numpy().copy(), d.cpu().numpy(), nxt.cpu().numpy(), env.state32.cpu().numpy()))
    for x, y in zip(*res):
        np.testing.assert_array_equal(x, y)
    # and the draw is NumPy's: second choice() of the stream seeded with 9
    rs = np.random.RandomState(9)
    rs.choice(cap, size=B)
    np.testing.assert_array_equal(res[0][2], rs.choice(cap, size=B))
    assert res[0][0][3] > 0 and res[0][0][4] > 0
