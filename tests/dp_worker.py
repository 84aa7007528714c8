"""One rank of a data-parallel parity run (started by test_dp_gloo.py / test_gpu_dp.py).

    python dp_worker.py <mode> <rank> <world> <port> <out.npz> <n_updates> <batch>

mode `oracle`: the NumPy restatement computes the gradients of this rank's shard of the minibatch,
the flat gradient buckets go through rl_gan_amd's GradExchange (gloo, CPU tensors) and every rank
applies the summed gradients.  mode `engine`: the HIP path does the same on cuda:0 (all ranks share
the one GPU of the test box; gloo moves the buckets).  mode `learner`: every rank runs a `CuriousLearner` of its
own (its own environments, replay ring and index stream) for <n_updates> iterations at <batch> rows per rank and also
saves the weights it started from and the minibatch rows of every update.  Each rank saves its final weights.
RLGAN_CHAINS in the environment selects the form of the multi-rank update (1 fused phases, 2 two chains).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def shard(batch, rank, world):
    per = batch.shape[0] // world
    return batch[rank * per:(rank + 1) * per]


def run_oracle(rank, world, n_updates, B):
    from gpu_util import make_batch
    from oracle.agent import CuriousAgent
    from rl_gan_amd.engine import GradExchange
    ex = GradExchange()
    assert ex.world == world and ex.rank == rank
    ag = CuriousAgent(seed=77, dtype=np.float64)

    def summed(*grads):
        bucket = torch.from_numpy(np.concatenate(grads))
        ex.all_reduce_sum(bucket)
        out, p = [], 0
        for g in grads:
            out.append(bucket[p:p + g.size].numpy().copy())
            p += g.size
        return out

    losses = []
    for u in range(n_updates):
        b = shard(make_batch(B, seed=u), rank, world)
        s, s_, g, a, cr = b[:, :7], b[:, 7:14], b[:, 14:21], b[:, 21:23], b[:, 24:25]
        c = ag.critic_grads(s, a, cr, s_, global_batch=B)
        gc, ga, lc = summed(c["g_critic"], c["g_actor"], np.array([c["loss"]]))
        ag.critic_apply(gc, ga)
        (ga2,) = summed(ag.actor_grads(s)["g_actor"])            # batch SUM: no scaling (DDPG.py:109)
        ag.actor_apply(ga2)
        d = ag.disc_grads(g, s_, s, a, global_batch=B)
        gd, ld = summed(d["g_disc"], np.array([d["loss"]]))
        ag.disc_apply(gd)
        gg = ag.gen_grads(s, a, global_batch=B)
        ggs, lg = summed(gg["g_gen"], np.array([gg["loss"]]))
        ag.gen_apply(ggs)
        losses.append([lc[0], ld[0], lg[0]])
    return {k: ag.flat(k) for k in ("actor_e", "actor_t", "critic_e", "critic_t", "gen", "disc")}, np.array(losses)


def run_engine(rank, world, n_updates, B):
    from gpu_util import Pair, dev, make_batch
    p = Pair(seed=77)
    assert p.eng.exchange.world == world
    losses = []
    for u in range(n_updates):
        b = dev(shard(make_batch(B, seed=u), rank, world))
        p.eng.update(p.A, p.C, p.G, p.D, b)
        l = p.eng.losses.clone()
        p.eng.exchange.all_reduce_sum(l)
        l = l.cpu().numpy()
        losses.append([l[p.eng.LOSS_CRITIC], l[p.eng.LOSS_D_STEP], l[p.eng.LOSS_G_STEP], l[p.eng.LOSS_D_EVAL], l[p.eng.LOSS_G_EVAL]])
    w = {"actor_e": p.A.e.get_flat(), "actor_t": p.A.t.get_flat(), "critic_e": p.C.e.get_flat(),
         "critic_t": p.C.t.get_flat(), "gen": p.G.net.get_flat(), "disc": p.D.net.get_flat()}
    return {k: np.asarray(v) for k, v in w.items()}, np.array(losses)


def run_learner(rank, world, n_updates, B):
    import rl_gan_amd as R
    torch.cuda.set_device(0)
    cfg = R.LearnerConfig(n_env=B, batch_size=B, memory_capacity=4 * B, seed=9, max_ep_steps=6, replace_iter_a=2, replace_iter_c=3)
    L = R.CuriousLearner(cfg)
    assert L.exchange.world == world and L.exchange.rank == rank
    nets = lambda: {"actor_e": L.actor.e.get_flat(), "actor_t": L.actor.t.get_flat(), "critic_e": L.critic.e.get_flat(),
                    "critic_t": L.critic.t.get_flat(), "gen": L.gen.net.get_flat(), "disc": L.disc.net.get_flat()}
    w0 = {"init_" + k: v for k, v in nets().items()}
    L.fill()
    losses, batches = [], []
    for _ in range(n_updates):
        batches.append(L.memory._data[L.idx][:, :25].cpu().numpy())     # the rows the coming update trains on
        L.step()
        l = L.losses()                                                   # a collective: global means
        losses.append([l["critic"], l["D_step"], l["G_step"], l["D_loss"], l["G_loss"]])
    out = {k: np.asarray(v) for k, v in nets().items()}
    out.update(w0)
    out["batches"] = np.stack(batches)
    out["idx_seed"] = np.array([cfg.seed + 7919 * rank])
    return out, np.array(losses)


def main():
    mode, rank, world, port, out, n_updates, B = sys.argv[1], *map(int, sys.argv[2:5]), sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    w, losses = {"oracle": run_oracle, "engine": run_engine, "learner": run_learner}[mode](rank, world, n_updates, B)
    np.savez(out, losses=losses, **w)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
