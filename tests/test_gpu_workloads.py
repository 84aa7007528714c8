"""The two pixel configurations bench.py times after its headline workload (tools/workloads.py) build and step at a small
batch: a workload that raised would show up in the bench line as an error string instead of a number."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_other_workloads_build_and_step():
    import workloads
    out = workloads.run_all(B=48, seconds=0.05)
    assert [o["workload"] for o in out] == ["conv_ddpg_doom_84_b4096", "pixel_curious_gan_universe_42_b4096"]
    for o in out:
        assert "error" not in o, o
        assert o["value"] > 0 and o["unit"] == "updates/s" and o["steps"] >= 5
