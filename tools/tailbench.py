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
ev_nosum = L.engine.eval_args_after_update(L.gen, L.disc, B, totals=False)
act = L._act_args()                                   # counters handed over between two copies, no ticket
act_tk = L._act_args(); act_tk.counters_out = None; act_tk.var_out = None      # advanced in place by the last workgroup
act_plain = L._act_args(); act_plain.counters = None; act_plain.counters_out = None; act_plain.var_out = None; act_plain.var_decay = 0.0
smp = L._sample_args()
run = lambda a, e, s, t: (lambda: check(lib.rlgan_rollout_eval_rows(a, e, s, t, _stream())))
for name, fn in [("eval, per-tile sums only", run(None, ev_nosum, None, None)),
                 ("eval + totals (ticket)", run(None, ev, None, tk)),
                 ("act, no counters", run(act_plain, None, None, None)),
                 ("act, counters in place (ticket)", run(act_tk, None, None, tk)),
                 ("act, counters handed over", run(act, None, None, None)),
                 ("draw only", run(None, None, smp, None)),
                 ("act + draw", run(act, None, smp, None)),
                 ("act + eval sums", run(act, ev_nosum, None, None)),
                 ("act + eval sums + draw", run(act, ev_nosum, smp, None)),
                 ("act + eval totals + draw (ticket)", run(act_tk, ev, smp, tk))]:
    print(f"{name:36s} {timeit(fn):8.1f} us")
