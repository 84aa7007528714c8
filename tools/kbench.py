#!/usr/bin/env python3
"""Times each hot-path kernel in isolation at batch 4096 (back-to-back launches of one kernel
bracketed by events on the launch stream) and prints achieved TFLOP/s against the fp32 peak.
A development aid; the judged numbers come from bench.py."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import rl_gan_amd as R
from rl_gan_amd.core import weight_grads
from bench import MACS_PER_ROW, PEAK_FP32_TFLOPS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REP = 50


def timeit(fn):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(REP):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REP * 1e3      # us


def timeit_graph(fn, reps=20):
    """Like timeit for kernels shorter than a launch from Python: `reps` calls recorded as one graph, the graph replayed."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        graph.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (10 * reps) * 1e3      # us


def main():
    cfg = R.LearnerConfig(n_env=B, batch_size=B, memory_capacity=max(1 << 16, 4 * B), seed=1)
    L = R.CuriousLearner(cfg)
    L.fill()
    L.update()
    e, A, C, G, D = L.engine, L.actor, L.critic, L.gen, L.disc
    bt = L.memory.gather(L.idx)
    s, s_, g, a, r = bt[:, :7], bt[:, 7:14], bt[:, 14:21], bt[:, 21:23], bt[:, 24]
    rows = {}
    rows["critic_step_rows"] = timeit(lambda: e.critic_rows(C, A, s, a, r, s_))
    rows["actor_step_rows"] = timeit(lambda: e.actor_rows(A, C, s))
    rows["disc_step_rows"] = timeit(lambda: e.disc_rows(D, g, s_, s, a))
    rows["gen_step_rows"] = timeit(lambda: e.gen_rows(G, D, s, a))
    rows["gan_eval_rows"] = timeit(lambda: e.gan_eval(G, D, g, s_, s, a)) - 0.0
    for k, us in rows.items():
        tf = 2.0 * MACS_PER_ROW[k] * B / (us * 1e-6) / 1e12
        print(f"{k:22s} {us:8.1f} us  {tf:6.1f} TFLOP/s  {100 * tf / PEAK_FP32_TFLOPS:5.1f}% of fp32 peak")
    wc = e.workspace("c", C.e, B)
    wd = e.workspace("d", D.net, 2 * B)
    wg = e.workspace("g", G.net, B)
    for name, net, ws in (("wgrad critic", C.e, wc), ("wgrad disc(2B rows)", D.net, wd), ("wgrad gen", G.net, wg)):
        us = timeit(lambda: weight_grads(net, ws))
        tf = 2.0 * net.n_params * ws.rows / (us * 1e-6) / 1e12
        print(f"{name:22s} {us:8.1f} us  {tf:6.1f} TFLOP/s  (wpass + reduce)")
    from rl_gan_amd._lib import check, lib
    from rl_gan_amd.core import _stream
    ca = e._critic_args(C, A, s, a, r, s_)[0]; da = e._disc_args(D, g, s_, s, a, 0.0)[0]
    aa = e._actor_args(A, C, s)[0]; ga = e._gen_args(G, D, s, a)[0]
    for name, fn in (("phase1_rows", lambda: check(lib.rlgan_phase1_rows(ca, da, _stream()))),
                     ("phase2_rows", lambda: check(lib.rlgan_phase2_rows(aa, ga, _stream())))):
        us = timeit(fn)
        tf = 2.0 * MACS_PER_ROW[name] * B / (us * 1e-6) / 1e12
        print(f"{name:22s} {us:8.1f} us  {tf:6.1f} TFLOP/s  {100 * tf / PEAK_FP32_TFLOPS:5.1f}% of fp32 peak")
    print(f"{'act step + draw':22s} {timeit(L.rollout_step):8.1f} us  (one launch: noise, act, env, store, cut, indices)")
    print(f"{'losses + act + draw':22s} {timeit(L._tail_body):8.1f} us  (one launch)")
    print(f"{'update (eager)':22s} {timeit(L._update_body):8.1f} us")
    L.capture()
    print(f"{'full step (graph)':22s} {timeit(L.step):8.1f} us")


if __name__ == "__main__":
    main()
