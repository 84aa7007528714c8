"""The pixel configurations bench.py times after its headline workload (tools/workloads.py) build and step at a small
batch: a workload that raised would show up in the bench line as an error string instead of a number."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_other_workloads_build_and_step():
    import workloads
    out = workloads.run_all(B=48, seconds=0.05)
    assert [o["workload"] for o in out] == ["conv_ddpg_doom_84_b4096", "conv_universe_84_b4096",
                                            "pixel_curious_gan_universe_42_b4096", "pixel_curious_gan_universe_42_b32",
                                            "conv_universe_84_b32"]
    for o in out:
        assert "error" not in o, o
        assert o["value"] > 0 and o["unit"] == "updates/s" and o["steps"] >= 5
        r = o["roofline"]
        assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and r["algorithmic_gflop"] > 0
    assert out[0]["batch"] == 48 and out[3]["batch"] == 32 and out[4]["batch"] == 32


def test_workload_multiply_accumulate_counts():
    """The per-frame counts the rooflines are priced with, against the layer shapes of the reference written out by
    hand (Conv_model.py:15-22,36-56: valid padding, so 84 -> 20 -> 9 -> 4 -> 1 and 84 -> 41 -> 20)."""
    import workloads as w
    per, shape = w.conv_macs(84, 84, 4, w.doom_head())
    assert per == [20 * 20 * 8 * 100, 9 * 9 * 16 * 72, 4 * 4 * 32 * 144, 64 * 288] and shape == (1, 1, 64) and sum(per) == 505472
    per, shape = w.conv_macs(84, 84, 4, [(32, 3, 2)] * 2)
    assert per == [1936512, 3686400] and shape == (20, 20, 32)
    assert w.encoder_update_macs(84, [(32, 3, 2)] * 2) == 2 * 8899712 + w.ddpg_step_macs(256, 6)
    per, shape = w.conv_macs(42, 42, 4, [(32, 3, 2)] * 2)
    assert sum(per) + 9 * 9 * 32 * 256 == 1870848


def _bench_line(extra_args, env=None, timeout=420):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_bench_starts_its_own_ranks_and_prints_one_line():
    """`python bench.py --gpus N` with no launcher around it starts its N ranks itself; with one GPU that start is
    rehearsed with --spawn, and RLGAN_FORCE_EXCHANGE=1 keeps the all-reduces (eager, captured, and in the two-chain form)
    in the one-rank step, so every multi-rank stage of the bench runs."""
    line, err = _bench_line(["--gpus", "1", "--steps", "5", "--warmup", "2", "--spawn"], env={"RLGAN_FORCE_EXCHANGE": "1"})
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["value"] > 0 and line["unit"] == "updates/s"
    assert line["config"]["launch"] == "graph" and line["eager_ms_per_step"] > 0
    assert set(line["update_forms"]) == {"fused_phases_ms_per_step", "two_chains_ms_per_step"}, line.get("update_forms")
    assert line["exchange_ms"] and 0 < line["roofline"]["frac"] < 1
    assert line["roofline"]["counters_from"]["measured_in_this_run"] is False


def test_bench_watchdog_keeps_the_headline_when_a_late_stage_hangs():
    """A stage limit too short for the two-chain form (it needs a graph capture and three timed regions) ends the run
    there: the line measured before it is printed, marked, and the exit status stays 0."""
    line, err = _bench_line(["--gpus", "1", "--steps", "5", "--warmup", "2", "--spawn"],
                            env={"RLGAN_FORCE_EXCHANGE": "1", "RLGAN_BENCH_HANG_STAGE": "two-chain form", "RLGAN_BENCH_STAGE_LIMIT": "20"})
    assert line["value"] > 0 and line["config"]["launch"] == "graph"
    assert line["update_forms"] == "two_chains timed out"
    assert "exceeded" in err
