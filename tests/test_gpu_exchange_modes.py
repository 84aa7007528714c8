"""The gradient exchange inside the captured training step, rehearsed with one rank (RLGAN_FORCE_EXCHANGE=1 keeps the
collective in the step): RCCL's all-reduce, in the fused-phase form and in the two-chain form of the update, must leave
the update untouched when there is nothing to add (one rank), eager and as a replayed HIP graph."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import rl_gan_amd as R
L = R.CuriousLearner(R.LearnerConfig(n_env=64, batch_size=64, memory_capacity=256, seed=3))
L.fill()
assert L.engine.exchange.enabled == (os.environ.get("RLGAN_FORCE_EXCHANGE") == "1")
for _ in range(3):
    L.step()
L.capture()
for _ in range(4):
    L.step()
out = dict(losses=L.losses(), chains=bool(L.engine.two_chains), actor=L.actor.e.get_flat().tolist()[:64], gen=L.gen.net.get_flat().tolist()[:64])
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
""" % ROOT


def _run(port, **env):
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", **env)
    r = subprocess.run([sys.executable, "-c", WORKER], env=e, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_exchange_forms_leave_a_one_rank_update_untouched():
    plain = _run(29551)
    rccl = _run(29552, RLGAN_FORCE_EXCHANGE="1")
    chains = _run(29554, RLGAN_FORCE_EXCHANGE="1", RLGAN_CHAINS="2")     # critic->actor and D->G as two chains on two streams
    assert chains["chains"] and not rccl["chains"]
    for other in (rccl, chains):
        assert other["losses"] == plain["losses"] and other["actor"] == plain["actor"] and other["gen"] == plain["gen"]
