#!/usr/bin/env python3
"""Throughput of the RL-GAN hot path on MI355X: learner updates/s and env-steps/s at batch 4096.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N bench.py ...`
(one rank per GPU, RCCL).  One "step" = one iteration of the training loop of the workload
`curious_gan_ddpg_arm_b4096`: an acting step of 4096 vectorised arm environments (actor, noise, G,
env.step, D, intrinsic reward, replay store) followed by one minibatch update at batch 4096 (index
draw, replay gather, critic/actor/D/G steps and the two post-update losses).  Per-GPU work is
fixed (weak scaling): every rank owns 4096 environments, a 2^20-row replay ring and a 4096-row
minibatch; gradients are summed over ranks.  Data are synthetic (random-initialised networks,
procedurally generated environments); all inputs are resident in HBM before the timed region.

Rank 0 prints one JSON line.  `other_workloads` (one GPU only) holds the reference's two pixel configurations
at the same batch, timed after the headline region and never part of `value`.  `roofline` describes the dominant kernel (timed live with events on
the launch stream); `cpu_baseline` is the NumPy float32 restatement of the same update timed on the
host cores of this box (a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENV = 4096
BATCH = 4096
CAPACITY = 1 << 20

# algorithmic multiply-accumulates per minibatch row of each kernel (forward + dX chains of the
# networks it walks; layer shapes 7/9/16 -> 200 -> 200 -> 10/50 -> 2/1/7)
A, C, G, D = 43420, 43810, 52150, 45210
MACS_PER_ROW = {
    "critic_step_rows": 2 * A + 3 * C + 42020,          # a'(s'), q'(s',a'), q(s,a), dX critic, actor fwd, dX actor
    "actor_step_rows": A + 2 * C + 42020,
    "disc_step_rows": 2 * (D + 42010),
    "gen_step_rows": G + 2 * D + 50350,
    "gan_eval_rows": 3 * D + G,
    "rollout_eval_rows": 3 * D + G + A + G + D,       # losses of the update + acting step, one launch
}
MACS_PER_ROW["phase1_rows"] = MACS_PER_ROW["critic_step_rows"] + MACS_PER_ROW["disc_step_rows"]   # one launch, both steps
MACS_PER_ROW["phase2_rows"] = MACS_PER_ROW["actor_step_rows"] + MACS_PER_ROW["gen_step_rows"]
PEAK_FP32_TFLOPS = 157.3   # MI355X dense fp32 matrix/vector peak


def measured_hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed counter run (profiles/*_pmc_hbm.json,
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command); None when there is none."""
    import glob
    import re
    # newest = highest (round, version) in the file name r<round>_v<version>_pmc_hbm.json, compared as numbers
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_v*_pmc_hbm.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return None
    with open(files[-1]) as f:
        k = json.load(f).get("kernels", {}).get(kernel.replace("_rows", "_kernel"))
    return k["hbm_bytes_per_launch"] if k else None


def measured_iteration_traffic():
    """HBM bytes of one whole iteration (row passes, weight gradients, optimizers, tail) from the same counter file."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_v*_pmc_hbm.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    return d.get("iteration_hbm_bytes")


def measured_mfma_busy(kernel):
    """Share of the matrix pipe's peak-clock cycles that held an MFMA during `kernel`, from the newest committed
    counter run (profiles/*_pmc_mfma.json, rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... of this same command)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_v*_pmc_mfma.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return None
    with open(files[-1]) as f:
        k = json.load(f).get("kernels", {}).get(kernel.replace("_rows", "_kernel"))
    return k.get("mfma_busy_frac") if k else None


def cpu_baseline(budget_s=12.0):
    """The oracle's float32 update at batch 4096 on the host cores (NumPy + its BLAS threads)."""
    import numpy as np
    from oracle.agent import CuriousAgent
    ag = CuriousAgent(seed=1234, dtype=np.float32)
    rng = np.random.RandomState(0)
    batch = rng.uniform(-1, 1, (BATCH, 25)).astype(np.float32)
    ag.update(batch)                                   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        ag.update(batch)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 200:
            break
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    out = {"value": round(n / dt, 3), "unit": "updates/s", "cores": threads, "kind": "port",
           "sample": f"{n} float32 NumPy updates at batch {BATCH} (critic, actor, D, G steps + 2 loss evals), {dt:.1f} s"}
    try:                                               # the same on one BLAS thread, a few seconds
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            ag.update(batch)
            n1, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < 4.0:
                ag.update(batch)
                n1 += 1
            out["single_thread_value"] = round(n1 / (time.perf_counter() - t1), 3)
    except Exception:                                  # threadpoolctl missing or no controllable BLAS: all-core figure only
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the two pixel configurations timed after the headline workload (one GPU only)")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying a HIP graph")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: 4096 environments and a global batch of 4096 split over the ranks (default: 4096 each per GPU)")
    args = ap.parse_args()

    # Native libraries write to file descriptor 1 as well (RCCL prints a version banner when its first
    # communicator comes up): keep the real stdout aside for the one JSON line and send the rest to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import rl_gan_amd as R
    n_env, batch = (N_ENV // world, BATCH // world) if args.strong else (N_ENV, BATCH)
    assert n_env * world == N_ENV or not args.strong, "--strong needs a rank count that divides 4096"
    cfg = R.LearnerConfig(n_env=n_env, batch_size=batch, memory_capacity=CAPACITY, seed=1)
    L = R.CuriousLearner(cfg)
    L.fill()                                            # 257 acting steps: the ring must be full before learning
    launch = "eager"
    if not args.eager:
        try:
            L.capture()                                 # one iteration = one HIP graph replay
            launch = "graph"
        except Exception as e:                          # keep the run alive: same kernels, launched one by one
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
            L._graph = None
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        L.step()

    # a one-rank job started through the launcher (RANK set) takes the multi-rank code paths below as well, which is how
    # they are rehearsed on a single GPU (with RLGAN_FORCE_EXCHANGE=1 the all-reduces stay in the step too)
    multi = dist.is_initialized()

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(steps):
        """EXACTLY `steps` iterations between two barrier + synchronize pairs; seconds, the maximum over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            L.step()
        barrier()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def measure(steps):
        """A short region (--steps 20 is under 4 ms) cannot raise the clocks by itself and one host hiccup is a large
        share of it: besides the --warmup steps the loop first runs for a fixed 0.3 s, then the region of exactly
        `steps` steps is timed `repeats` times back to back and the MEDIAN region is the one reported."""
        if not multi:
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                L.step()
            torch.cuda.synchronize()
        else:
            # every step holds collectives, so all ranks must take the same number of steps: time 50 of them (maximum
            # over ranks, the same value everywhere) and derive the count for 0.3 s from that
            per_step = timed_region(50) / 50
            for _ in range(min(20000, int(0.3 / max(per_step, 1e-6)) + 1)):
                L.step()
        repeats = max(3, min(25, int(round(0.25 / (steps * 2e-4)))) | 1)     # odd; about a quarter of a second in all
        regions = sorted(timed_region(steps) for _ in range(repeats))
        return regions[len(regions) // 2], regions[0], regions[-1], repeats

    dt, dt_min, dt_max, repeats = measure(args.steps)
    forms = None
    if L.engine.exchange.enabled and launch == "graph":
        # the other form of the multi-rank update (two chains on two streams, one exchange per step), same run, same
        # measurement: the scaling record then shows which form the exchange latency favours
        try:
            L.engine.set_chains(2)
            L._graph = None
            L.capture()
            dt2 = measure(args.steps)[0]
            forms = {"fused_phases_ms_per_step": round(dt / args.steps * 1e3, 4), "two_chains_ms_per_step": round(dt2 / args.steps * 1e3, 4)}
        except Exception as e:
            print(f"[bench] two-chain form not timed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
        L.engine.set_chains(1)

    # per-kernel durations, measured live on the launch stream over a second, eager pass of the same loop
    L._graph = None
    L.engine.kernel_events = {}
    for _ in range(min(args.steps, 50)):
        L.step()
    times = L.engine.kernel_times_ms()
    L.engine.kernel_events = None
    losses = L.losses()
    # the acting path alone (no learning): vector steps of all environments, one launch each
    barrier()
    ta = time.perf_counter()
    for _ in range(200):
        L.rollout_step()
    barrier()
    acting = world * n_env * 200 / (time.perf_counter() - ta)

    if rank == 0:
        # median over the eager pass: robust against the first launches (lazy work-array setup)
        avg_ms = {k: sorted(v)[len(v) // 2] for k, v in times.items() if k in MACS_PER_ROW}
        # with more than one rank: the gradient all-reduces of the two phases (name = bucket size), same event timing
        exchange_ms = {k: sorted(v)[len(v) // 2] for k, v in times.items() if k.startswith("exchange_")}
        dom = max(avg_ms, key=avg_ms.get)
        flop = 2.0 * MACS_PER_ROW[dom] * batch
        achieved = flop / (avg_ms[dom] * 1e-3) / 1e12
        out = {
            "metric": "learner updates/sec (batch 4096)", "value": round((1 if args.strong else world) * args.steps / dt, 2), "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "repeats": repeats, "ms_per_step_min": round(dt_min / args.steps * 1e3, 4), "ms_per_step_max": round(dt_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "curious_gan_ddpg_arm_b4096", "envs_per_gpu": n_env, "batch_per_gpu": batch,
                       "replay_rows_per_gpu": CAPACITY, "global_batch": batch * world, "parallelism": f"dp{world}", "launch": launch},
            "env_steps_per_s": round(world * n_env * args.steps / dt, 1),
            "acting_only_env_steps_per_s": round(acting, 1),
            "kernel_ms": {k: round(v, 4) for k, v in sorted(avg_ms.items())},
            "losses": {k: round(v, 6) for k, v in losses.items()},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": measured_hbm_traffic(dom),
                         "mfma_busy": measured_mfma_busy(dom), "traffic_per_iteration": measured_iteration_traffic()},
        }
        if forms:
            out["update_forms"] = forms
        if exchange_ms:
            out["exchange_ms"] = {k: round(v, 4) for k, v in sorted(exchange_ms.items())}
        if world == 1 and not multi and not args.no_other_workloads:
            # the reference's two pixel configurations at the same batch, a couple of seconds each (tools/workloads.py);
            # reported beside the headline, never part of `value`
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from workloads import run_all
            del L
            torch.cuda.empty_cache()
            out["other_workloads"] = run_all(BATCH)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
