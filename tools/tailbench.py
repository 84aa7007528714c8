#!/usr/bin/env python3
"""Timing of the rollout/eval launch in its variants (which part costs what)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rl_gan_amd as R
from rl_gan_amd import _lib
from rl_gan_amd._lib import check, lib
from rl_gan_amd.core import _stream
from kbench import timeit

B = 4096
cfg = R.LearnerConfig(n_env=B, batch_size=B, memory_capacity=1 << 16, seed=1)
L = R.CuriousLearner(cfg)
L.fill()
L.step(); L.step()
tk = L.engine.ticket().data_ptr()
ev = L.engine.eval_args_after_update(L.gen, L.disc, B)
ev_nosum = L.engine.eval_args_after_update(L.gen, L.disc, B); ev_nosum.loss_out = None
act = L._act_args()
act_plain = L._act_args(); act_plain.counters = None; act_plain.var_decay = 0.0
smp = L._sample_args()
run = lambda a, e, s, t: (lambda: check(lib.rlgan_rollout_eval_rows(a, e, s, t, _stream())))
for name, fn in [("eval, no totals (no ticket)", run(None, ev_nosum, None, None)),
                 ("eval + totals (ticket)", run(None, ev, None, tk)),
                 ("act plain (no ticket)", run(act_plain, None, None, None)),
                 ("act + counters (ticket)", run(act, None, None, tk)),
                 ("draw only", run(None, None, smp, None)),
                 ("act plain + draw", run(act_plain, None, smp, None)),
                 ("act plain + eval no totals", run(act_plain, ev_nosum, None, None)),
                 ("act + eval + draw (ticket)", run(act, ev, smp, tk))]:
    print(f"{name:32s} {timeit(fn):8.1f} us")
