"""Data-parallel HIP path, two ranks sharing the one GPU of the test box (gloo carries the gradient
buckets; on a multi-GPU node the same code runs one rank per GPU over RCCL): each rank updates on
half of a minibatch and must end with the weights the oracle gets from the whole minibatch."""
import numpy as np
import pytest

from gpu_util import Pair, close, make_batch
from test_dp_gloo import KEYS, run_ranks

pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_gpu_match_the_oracle(tmp_path):
    n_updates, B = 3, 256
    ranks = run_ranks("engine", 2, n_updates, B, tmp_path)
    p = Pair(seed=77)
    losses = []
    for u in range(n_updates):
        out = p.o.update(make_batch(B, seed=u))
        losses.append([out["critic_loss"], out["disc"]["loss"], out["gen"]["loss"], out["D_loss"], out["G_loss"]])
    for k in KEYS:
        assert np.array_equal(ranks[0][k], ranks[1][k]), f"replicas diverged in {k}"
        atol = 0.25 * n_updates * 1e-4 if k in ("gen", "disc") else 0.0       # see gpu_util.Pair.check_weights
        close(ranks[0][k], p.o.flat(k), what=k, atol=atol)
    close(ranks[0]["losses"], np.array(losses), what="losses")
