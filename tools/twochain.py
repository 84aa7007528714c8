#!/usr/bin/env python3
"""Experiment: the DDPG chain (critic step -> actor step) and the GAN chain (D step -> G step) of one
update on two streams inside one graph, against the two fused phase launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rl_gan_amd as R
from rl_gan_amd.core import launch_wgrads, launch_apply, wgrad_job, reduce_job, adam_job
from kbench import timeit

B = 4096
L = R.CuriousLearner(R.LearnerConfig(n_env=B, batch_size=B, memory_capacity=1 << 16, seed=1))
L.fill()
L.step(); L.step()
e, A, C, G, D = L.engine, L.actor, L.critic, L.gen, L.disc
S, Ad = 7, 2
ring, idx = L.memory._data, L.idx
b_s, b_s_, b_g, b_a, b_r = ring[:, :S], ring[:, S:2 * S], ring[:, 2 * S:3 * S], ring[:, 3 * S:3 * S + Ad], ring[:, 3 * S + Ad + 1]
from rl_gan_amd._lib import lib, check
from rl_gan_amd.core import _stream

def ddpg_chain():
    cargs, wc, wa, lpc = e._critic_args(C, A, b_s, b_a, b_r, b_s_, idx)
    check(lib.rlgan_critic_step_rows(cargs, _stream()))
    wj, rj = e._critic_jobs(C, A, wc, wa, lpc, B)
    launch_wgrads(wj, S, Ad)
    launch_apply(e._critic_apply_jobs(C, A, wc, wa), rj)
    aargs, wa2 = e._actor_args(A, C, b_s, idx)
    check(lib.rlgan_actor_step_rows(aargs, _stream()))
    launch_wgrads([wgrad_job(A.e, wa2)], S, Ad)
    launch_apply([e._actor_apply_job(A, wa2)], [reduce_job(A.e, wa2)])

def gan_chain():
    dargs, wd, lpd = e._disc_args(D, b_g, b_s_, b_s, b_a, 0.0, idx)
    check(lib.rlgan_disc_step_rows(dargs, _stream()))
    launch_wgrads([wgrad_job(D.net, wd)], S, Ad)
    launch_apply([adam_job(D.net, wd.grad, D.m, D.v, D.step, D.lr)],
                 [reduce_job(D.net, wd, lpd, lpd.numel(), e._inv_gb(B), e.losses[e.LOSS_D_STEP:])])
    gargs, wg, lpg = e._gen_args(G, D, b_s, b_a, idx)
    check(lib.rlgan_gen_step_rows(gargs, _stream()))
    launch_wgrads([wgrad_job(G.net, wg)], S, Ad)
    launch_apply([adam_job(G.net, wg.grad, G.m, G.v, G.step, G.lr)],
                 [reduce_job(G.net, wg, lpg, lpg.numel(), e._inv_gb(B), e.losses[e.LOSS_G_STEP:])])

s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        gan_chain()
    ddpg_chain()
    cur.wait_stream(s1)

def graph_of(fn):
    with torch.cuda.stream(s0):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s0):
        fn()
    return g

# fresh workspaces for the per-step grads (the fused path rebinds .grad to bucket slices)
for tag, net, rows in (("c", C.e, B), ("ca", A.e, B), ("a", A.e, B), ("d", D.net, 2 * B), ("g", G.net, B)):
    ws = e.workspace(tag, net, rows)
    ws.grad = torch.zeros(net.n_params, device="cuda")
g_fused = graph_of(lambda: e.update(A, C, G, D, ring, 0.0, with_eval=False, idx=idx))
g_one = graph_of(lambda: (ddpg_chain(), gan_chain()))
g_two = graph_of(two_streams)
for name, g in (("fused phases, one stream", g_fused), ("separate steps, one stream", g_one), ("two chains on two streams", g_two)):
    print(f"{name:32s} {timeit(g.replay):8.1f} us")
