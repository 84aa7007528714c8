"""Stand-in rank for tests/test_bench_launch.py: what a rank of bench.py does around its GPU work, over gloo on the CPU.

    python spawn_worker.py ok            every rank all-reduces its rank number, rank 0 prints one JSON line
    python spawn_worker.py fail          rank 1 leaves with status 7 before the collective
    python spawn_worker.py hang          a fallback line is registered, then the stage never ends: the watchdog prints it
    python spawn_worker.py hang-early    the stage hangs before any line exists: status 3
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    dog = bench.Watchdog(rank, json_fd)
    if mode == "fail" and rank == 1:
        sys.exit(7)
    if mode == "hang-early":
        dog.arm(1.0, "set-up")
        time.sleep(60)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    line = {"metric": "stand-in", "value": float(t.item()), "n_gpus": world}
    if mode == "hang":
        dog.fallback, dog.have_line = line, True
        dog.arm(1.0, "two-chain form", mark={"update_forms": "two_chains timed out"})
        time.sleep(60)
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
