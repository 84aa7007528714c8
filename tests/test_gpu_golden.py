"""The HIP path against the committed known answers of the network half (tests/golden/nn_*.npz):
losses, sampled gradient entries, intrinsic rewards and final weights after three updates at the
reference's own batch sizes (16: Curious_GAN.py:333; 32: the DDPG-only configuration)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_curious_update_b16_matches_fixture():
    from gpu_util import Pair, close, dev
    from oracle_util import NETS
    fx = np.load(os.path.join(GOLDEN, "nn_update_b16.npz"))
    p = Pair(seed=1234)
    eng = p.eng
    # acting path quantities under the initial weights
    from rl_gan_amd.core import mlp_forward
    s, a = dev(fx["ir_s"]), dev(fx["ir_a"])
    act = mlp_forward(p.A.e, s)
    close(act.cpu().numpy(), fx["ir_action"], what="actions")
    g = mlp_forward(p.G.net, s, a=a)
    close(g.cpu().numpy(), fx["ir_g"], what="G(s,a)")
    d = mlp_forward(p.D.net, s, x=g, a=a, out_mode=2)          # 1 - D(s, a, g)
    cur = fx["ir_r"].astype(np.float64) + 0.01 * d.cpu().numpy().ravel().astype(np.float64)
    close(cur, fx["ir_curious_r"], what="intrinsic reward")
    for i, b in enumerate(fx["batch"]):
        bt = dev(b)
        if i == 0:
            # step by step, to see every gradient
            S = dict(s=bt[:, :7], s_=bt[:, 7:14], g=bt[:, 14:21], a=bt[:, 21:23], cr=bt[:, 24])
            wc, wa = eng.critic_step(p.C, p.A, S["s"], S["a"], S["cr"], S["s_"])
            got = {"g_critic": wc.grad.cpu().numpy(), "g_actor_from_critic": wa.grad.cpu().numpy()}
            got["g_actor"] = eng.actor_step(p.A, p.C, S["s"]).grad.cpu().numpy()
            got["g_disc"] = eng.disc_step(p.D, S["g"], S["s_"], S["s"], S["a"]).grad.cpu().numpy()
            got["g_gen"] = eng.gen_step(p.G, p.D, S["s"], S["a"]).grad.cpu().numpy()
            eng.gan_eval(p.G, p.D, S["g"], S["s_"], S["s"], S["a"])
            for k, v in got.items():
                stride = int(round(v.size / fx[k].size))
                # tolerance scale = the whole tensor's, as in the live comparisons
                scale = np.sqrt(fx[k + "_sums"][2] / v.size)
                close(v[::stride], fx[k], what=k, atol=1e-5 * scale)
                close(v.astype(np.float64).sum(), fx[k + "_sums"][0], what=k + " sum", atol=1e-5 * fx[k + "_sums"][1])
        else:
            eng.update(p.A, p.C, p.G, p.D, bt)
        l = eng.losses.cpu().numpy()
        close(l[[eng.LOSS_CRITIC, eng.LOSS_D_STEP, eng.LOSS_G_STEP, eng.LOSS_D_EVAL, eng.LOSS_G_EVAL]], fx["losses"][i], what=f"losses {i}")
    w = {"actor_e": p.A.e, "actor_t": p.A.t, "critic_e": p.C.e, "critic_t": p.C.t, "gen": p.G.net, "disc": p.D.net}
    for k in NETS:
        v = w[k].get_flat()
        stride = int(round(v.size / fx[f"final_{k}"].size))
        # Adam-trained weights: plus the fixture's per-entry bound on what gradient noise does to them (about a step
        # where an entry's gradient is rounding noise, next to nothing elsewhere; oracle/optim.py, Adam.allow)
        close(v[::stride], fx[f"final_{k}"], what=f"final {k}", atol=fx[f"final_allow_{k}"] if k in ("gen", "disc") else 0.0)


def test_ddpg_update_b32_matches_fixture():
    from gpu_util import Pair, close, dev
    from oracle_util import NETS
    fx = np.load(os.path.join(GOLDEN, "nn_ddpg_b32.npz"))
    p = Pair(seed=4321)
    for i, b in enumerate(fx["batch"]):
        p.eng.update_ddpg(p.A, p.C, dev(b))
        close(p.eng.losses[p.eng.LOSS_CRITIC].item(), fx["losses"][i][0], what=f"critic loss {i}")
    w = {"actor_e": p.A.e, "actor_t": p.A.t, "critic_e": p.C.e, "critic_t": p.C.t}
    for k in w:
        v = w[k].get_flat()
        stride = int(round(v.size / fx[f"final_{k}"].size))
        close(v[::stride], fx[f"final_{k}"], what=f"final {k}")
