"""Data-parallel contract on CPU, world_size 2 over gloo: gradients of mean-type losses are divided
by the GLOBAL batch on every rank, the actor's policy gradient is an unscaled batch sum, buckets are
summed over ranks, every rank applies the same update.  Two ranks working on half a minibatch each
must end with the weights a single process gets from the whole minibatch (the oracle's `update`),
and the two replicas must stay bit-identical."""
import os
import socket
import subprocess
import sys

import numpy as np

from oracle.agent import CuriousAgent
from gpu_util import make_batch

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("actor_e", "actor_t", "critic_e", "critic_t", "gen", "disc")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(mode, world, n_updates, B, tmp_path, timeout=300, env=None):
    port = free_port()
    env = dict(os.environ, **(env or {}))
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), mode, str(r), str(world), str(port),
                               outs[r], str(n_updates), str(B)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return [np.load(o) for o in outs]


def single_process(n_updates, B):
    ag = CuriousAgent(seed=77, dtype=np.float64)
    losses = []
    for u in range(n_updates):
        out = ag.update(make_batch(B, seed=u))
        losses.append([out["critic_loss"], out["disc"]["loss"], out["gen"]["loss"]])
    return ag, np.array(losses)


def test_two_ranks_match_one_process(tmp_path):
    n_updates, B = 3, 64
    ranks = run_ranks("oracle", 2, n_updates, B, tmp_path)
    ag, losses = single_process(n_updates, B)
    for k in KEYS:
        assert np.array_equal(ranks[0][k], ranks[1][k]), f"replicas diverged in {k}"
        np.testing.assert_allclose(ranks[0][k], ag.flat(k), rtol=1e-9, atol=1e-12, err_msg=k)
    np.testing.assert_allclose(ranks[0]["losses"], losses, rtol=1e-10)
    assert np.array_equal(ranks[0]["losses"], ranks[1]["losses"])
