#!/usr/bin/env python3
"""Writes tests/golden/nn_update_b16.npz and nn_ddpg_b32.npz: known answers for the network half of
the hot path (update steps, losses, gradients, intrinsic rewards).

Parity unpinned: the reference computes these in TensorFlow 1.x, which cannot be installed here, and
its repository holds no test or stored output for them.  The numbers below come from the float64
NumPy restatement in oracle/ (whose gradients are cross-checked against torch autograd by
tests/test_oracle_autograd.py).  The fixtures freeze that restatement so that any later drift of the
oracle or of the HIP path shows up against committed data.

    python tests/golden/gen_nn_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle_util import NETS, make_batch, make_oracle  # noqa: E402


GRAD_STRIDE, FINAL_STRIDE = 4, 8     # every 4th gradient / 8th weight entry is stored, plus sums over all


def sums(v):
    v = np.asarray(v, dtype=np.float64)
    return np.array([v.sum(), np.abs(v).sum(), (v * v).sum()])


def run(seed, B, iters, ddpg_only):
    ag = make_oracle(seed)
    out = {}
    for k in NETS:                       # the initial weights are regenerated from the seed; pin their sums
        out[f"init_sums_{k}"] = sums(ag.flat(k))
    batches = np.stack([make_batch(B, seed=seed + 10 + i) for i in range(iters)])
    out["batch"] = batches.astype(np.float32)
    # intrinsic reward of 64 transitions under the initial weights (Curious_GAN.py:192-194)
    tr = make_batch(64, seed=seed + 99)
    s, a, r = tr[:, :7], tr[:, 21:23], tr[:, 23]
    g = ag.predict(s, a)
    out["ir_s"], out["ir_a"], out["ir_r"] = s.astype(np.float32), a.astype(np.float32), r.astype(np.float32)
    out["ir_g"] = g.astype(np.float32)
    out["ir_curious_r"] = ag.intrinsic_reward(r, s, a, g).ravel()
    out["ir_action"] = ag.choose_action(s)
    losses = []
    for i in range(iters):
        res = (ag.update_ddpg if ddpg_only else ag.update)(batches[i])
        if i == 0:
            grads = {"g_critic": res["critic"]["g_critic"], "g_actor_from_critic": res["critic"]["g_actor"],
                     "g_actor": res["actor"]["g_actor"]}
            if not ddpg_only:
                grads.update(g_disc=res["disc"]["g_disc"], g_gen=res["gen"]["g_gen"])
            for k, v in grads.items():
                out[k] = v[::GRAD_STRIDE].astype(np.float32)
                out[k + "_sums"] = sums(v)
        if ddpg_only:
            losses.append([res["critic_loss"]])
        else:
            losses.append([res["critic_loss"], res["disc"]["loss"], res["gen"]["loss"], res["D_loss"], res["G_loss"]])
    out["losses"] = np.array(losses)
    for k in NETS:
        out[f"final_{k}"] = ag.flat(k)[::FINAL_STRIDE].astype(np.float32)
        out[f"final_sums_{k}"] = sums(ag.flat(k))
    if not ddpg_only:
        # per-entry bound on what gradient errors within 1e-5 may do to the Adam-trained weights (oracle/optim.py)
        out["final_allow_gen"] = ag.opt_g.allow[::FINAL_STRIDE].astype(np.float32)
        out["final_allow_disc"] = ag.opt_d.allow[::FINAL_STRIDE].astype(np.float32)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "nn_update_b16.npz"), **run(1234, 16, 3, False))
    np.savez_compressed(os.path.join(HERE, "nn_ddpg_b32.npz"), **run(4321, 32, 3, True))
    for f in ("nn_update_b16.npz", "nn_ddpg_b32.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
