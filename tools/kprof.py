#!/usr/bin/env python3
"""Launches one hot-path kernel a few times at a given batch size: a small target for
`rocprofv3 --pmc` passes.  usage: kprof.py <kernel> [B] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl_gan_amd as R
from rl_gan_amd.core import weight_grads

which = sys.argv[1] if len(sys.argv) > 1 else "actor"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = R.CuriousLearner(R.LearnerConfig(n_env=B, batch_size=B, memory_capacity=max(1 << 16, 4 * B), seed=1))
L.fill()
L.update()
e, A, C, G, D = L.engine, L.actor, L.critic, L.gen, L.disc
bt = L.memory.gather(L.idx)
s, s_, g, a, r = bt[:, :7], bt[:, 7:14], bt[:, 14:21], bt[:, 21:23], bt[:, 24]
fn = {"critic": lambda: e.critic_rows(C, A, s, a, r, s_), "actor": lambda: e.actor_rows(A, C, s),
      "disc": lambda: e.disc_rows(D, g, s_, s, a), "gen": lambda: e.gen_rows(G, D, s, a),
      "eval": lambda: e.gan_eval(G, D, g, s_, s, a),
      "wgrad": lambda: weight_grads(C.e, e.workspace("c", C.e, B)), "act": L.rollout_step}[which]
torch.cuda.synchronize()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
