"""Data-parallel HIP path, two ranks sharing the one GPU of the test box (gloo carries the gradient
buckets; on a multi-GPU node the same code runs one rank per GPU over RCCL): each rank updates on
half of a minibatch and must end with the weights the oracle gets from the whole minibatch.  Both forms
of the multi-rank update run: the fused phases (the default) and the two chains (RLGAN_CHAINS=2)."""
import numpy as np
import pytest

from gpu_util import Pair, close, make_batch
from oracle_util import load_weights
from test_dp_gloo import KEYS, run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("chains", ["1", "2"])
def test_two_ranks_on_one_gpu_match_the_oracle(tmp_path, chains):
    n_updates, B = 3, 256
    ranks = run_ranks("engine", 2, n_updates, B, tmp_path, env={"RLGAN_CHAINS": chains})
    p = Pair(seed=77)
    losses = []
    for u in range(n_updates):
        out = p.o.update(make_batch(B, seed=u))
        losses.append([out["critic_loss"], out["disc"]["loss"], out["gen"]["loss"], out["D_loss"], out["G_loss"]])
    p.o_steps = n_updates
    for k in KEYS:
        assert np.array_equal(ranks[0][k], ranks[1][k]), f"replicas diverged in {k}"
    p.check_flats({k: ranks[0][k] for k in KEYS}, what=f"chains={chains}", adam_steps=n_updates)
    close(ranks[0]["losses"], np.array(losses), what="losses")


@pytest.mark.parametrize("chains", ["1", "2"])
def test_two_learners_match_one_process_on_the_joined_minibatches(tmp_path, chains):
    """`CuriousLearner` itself at world 2: every rank rolls out its own environments into its own ring and draws
    its own indices; the replicas must stay bit-identical, and their losses and weights must be those of ONE
    process (the oracle) that starts from the same weights and trains on the rank-ordered concatenation of the two
    minibatches of every update -- the contract of the data-parallel learner."""
    from oracle.agent import CuriousAgent, Hyper
    n_updates, B = 3, 48
    ranks = run_ranks("learner", 2, n_updates, B, tmp_path, env={"RLGAN_CHAINS": chains})
    for k in KEYS:
        assert np.array_equal(ranks[0][k], ranks[1][k]), f"replicas diverged in {k}"
        assert np.array_equal(ranks[0]["init_" + k], ranks[1]["init_" + k])
    assert not np.array_equal(ranks[0]["batches"], ranks[1]["batches"])          # per-rank data
    assert np.array_equal(ranks[0]["losses"], ranks[1]["losses"])
    ag = CuriousAgent(seed=0, hyper=Hyper(replace_iter_a=2, replace_iter_c=3))
    load_weights(ag, {k: ranks[0]["init_" + k] for k in KEYS})
    for u in range(n_updates):
        joined = np.concatenate([ranks[0]["batches"][u], ranks[1]["batches"][u]]).astype(np.float64)
        ref = ag.update(joined)
        got = ranks[0]["losses"][u]
        close(got[0], ref["critic_loss"], what=f"critic loss {u}")
        close(got[1], ref["disc"]["loss"], what=f"D step loss {u}")
        close(got[2], ref["gen"]["loss"], what=f"G step loss {u}")
        close(got[3], ref["D_loss"], what=f"D_loss {u}")
        close(got[4], ref["G_loss"], what=f"G_loss {u}")
    p = Pair.__new__(Pair)
    p.o, p.hp = ag, ag.hp
    p.check_flats({k: ranks[0][k] for k in KEYS}, what=f"learner chains={chains}", adam_steps=n_updates)
