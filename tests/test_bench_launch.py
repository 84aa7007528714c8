"""bench.py's launcher-less start (`--gpus N` with no RANK in the environment) and its stage watchdog, on the CPU: the
ranks are stand-in workers over gloo (tests/spawn_worker.py) that use bench.py's own `spawn_ranks` and `Watchdog`."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "spawn_worker.py")


def _spawn(mode, n=2):
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(%d, [%r], script=%r, need_gpus=False))" % (ROOT, n, mode, WORKER))
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    return p, time.time() - t0


def test_two_ranks_start_and_rank0_line_is_relayed():
    p, _ = _spawn("ok")
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"metric": "stand-in", "value": 1.0, "n_gpus": 2}


def test_failed_rank_ends_the_job_with_its_status():
    p, dt = _spawn("fail")
    assert p.returncode == 7
    assert "rank 1 ended with status 7" in p.stderr
    assert dt < 60                                  # rank 0 was waiting in the rendezvous: it has been stopped


def test_hung_stage_prints_the_line_known_so_far():
    p, dt = _spawn("hang")
    assert p.returncode == 0, p.stderr
    line = json.loads(p.stdout.strip())
    assert line["value"] == 1.0 and line["update_forms"] == "two_chains timed out"
    assert dt < 40


def test_hang_before_any_line_is_an_error():
    p, dt = _spawn("hang-early")
    assert p.returncode == 3
    assert p.stdout.strip() == ""
    assert dt < 40


def test_more_gpus_than_visible_is_refused_before_any_rank_starts():
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(n, 2))], capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() >= max(n, 2):
        return
    assert p.returncode == 2 and "GPU(s) are visible" in p.stderr
