"""The oracle against the committed known answers of the network half (tests/golden/nn_*.npz, made
by tests/golden/gen_nn_fixtures.py).  Parity unpinned: the reference holds no stored outputs for its
TensorFlow path, so these freeze the float64 restatement (cross-checked by torch autograd in
test_oracle_autograd.py) rather than pin it to the reference."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle_util import NETS, make_oracle

SUM_TOL = dict(rtol=1e-9, atol=1e-12)


def sums(v):
    v = np.asarray(v, dtype=np.float64)
    return np.array([v.sum(), np.abs(v).sum(), (v * v).sum()])


@pytest.mark.parametrize("name,seed,ddpg_only", [("nn_update_b16.npz", 1234, False), ("nn_ddpg_b32.npz", 4321, True)])
def test_oracle_reproduces_the_network_fixtures(name, seed, ddpg_only):
    fx = np.load(os.path.join(GOLDEN, name))
    ag = make_oracle(seed)
    for k in NETS:
        np.testing.assert_allclose(sums(ag.flat(k)), fx[f"init_sums_{k}"], **SUM_TOL, err_msg=f"init {k}")
    s, a, r = (fx[k].astype(np.float64) for k in ("ir_s", "ir_a", "ir_r"))
    g = ag.predict(s, a)
    np.testing.assert_allclose(g, fx["ir_g"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ag.intrinsic_reward(r, s, a, g).ravel(), fx["ir_curious_r"], rtol=1e-12)
    np.testing.assert_allclose(ag.choose_action(s), fx["ir_action"], rtol=1e-12, atol=1e-15)
    gs, fs = int(round(ag.flat("actor_e").size / fx["final_actor_e"].size)), None
    for i, b in enumerate(fx["batch"].astype(np.float64)):
        res = (ag.update_ddpg if ddpg_only else ag.update)(b)
        got = [res["critic_loss"]] if ddpg_only else [res["critic_loss"], res["disc"]["loss"], res["gen"]["loss"], res["D_loss"], res["G_loss"]]
        np.testing.assert_allclose(got, fx["losses"][i], rtol=1e-12)
        if i == 0:
            grads = {"g_critic": res["critic"]["g_critic"], "g_actor_from_critic": res["critic"]["g_actor"], "g_actor": res["actor"]["g_actor"]}
            if not ddpg_only:
                grads.update(g_disc=res["disc"]["g_disc"], g_gen=res["gen"]["g_gen"])
            for k, v in grads.items():
                np.testing.assert_allclose(sums(v), fx[k + "_sums"], **SUM_TOL, err_msg=k)
                stride = int(round(v.size / fx[k].size))
                np.testing.assert_allclose(v[::stride], fx[k], rtol=1e-6, atol=1e-12, err_msg=k)
    for k in NETS:
        np.testing.assert_allclose(sums(ag.flat(k)), fx[f"final_sums_{k}"], **SUM_TOL, err_msg=f"final {k}")
        np.testing.assert_allclose(ag.flat(k)[::gs], fx[f"final_{k}"], rtol=1e-6, atol=1e-9, err_msg=f"final {k}")
