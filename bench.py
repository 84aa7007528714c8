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


def _newest_counter_file(kind):
    """Newest committed counter summary profiles/r<round>_v<version>_pmc_<kind>.json (highest round, then version)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_v*_pmc_{kind}.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.basename(files[-1])


def committed_counters(kernel):
    """Counter-derived figures of `kernel` that are NOT measured in this run: they are read from the newest committed
    rocprofv3 --pmc summaries of this same command (FETCH_SIZE / WRITE_SIZE passes -> HBM-side bytes per launch and per
    iteration; SQ_VALU_MFMA_BUSY_CYCLES pass -> share of the matrix pipe's peak-clock cycles that held an MFMA).
    Returns (traffic, traffic_per_iteration, mfma_busy, source) with `source` naming the files and, when the summaries
    record it, the commit of the library they were taken on."""
    name = kernel.replace("_rows", "_kernel")
    hbm, hbm_file = _newest_counter_file("hbm")
    mfma, mfma_file = _newest_counter_file("mfma")
    traffic = per_iter = busy = None
    if hbm:
        k = hbm.get("kernels", {}).get(name)
        traffic = k["hbm_bytes_per_launch"] if k else None
        per_iter = hbm.get("iteration_hbm_bytes")
    if mfma:
        k = mfma.get("kernels", {}).get(name)
        busy = k.get("mfma_busy_frac") if k else None
    source = {"files": [f"profiles/{f}" for f in (hbm_file, mfma_file) if f], "measured_in_this_run": False}
    commit = (hbm or {}).get("commit") or (mfma or {}).get("commit")
    if commit:
        source["commit"] = commit
    return traffic, per_iter, busy, source


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


class Watchdog:
    """Wall-clock guard for the stages of a run.  `arm(seconds, stage)` starts (or restarts) the clock; `disarm()` stops
    it.  When a stage overruns, the process leaves at once with `os._exit` (a fresh exit: nothing is re-executed, no
    collective is waited for): rank 0 first prints the best JSON line known so far (`fallback`), marked with the stage
    that timed out; without such a line every rank leaves with status 3.  Every rank arms the same stages with the same
    limits, so the ranks of a hung collective all leave within moments of each other."""

    def __init__(self, rank, json_fd):
        import threading
        self.rank, self.json_fd = rank, json_fd
        self.fallback = None            # dict: the line to print if a later stage hangs
        self.have_line = False          # set on every rank once rank 0 holds a fallback line
        self._lock = threading.Lock()
        self._deadline, self._stage, self._limit = None, None, None
        self._mark = None               # stage -> key/value merged into the fallback line on expiry
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, seconds, stage, mark=None):
        with self._lock:
            self._deadline, self._stage, self._limit, self._mark = time.monotonic() + seconds, stage, seconds, mark
        if os.environ.get("RLGAN_BENCH_HANG_STAGE") == stage:      # fault injection for the tests: this stage never ends
            time.sleep(10 * seconds + 60)

    def disarm(self):
        with self._lock:
            self._deadline = None

    def _run(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                late = self._deadline is not None and time.monotonic() > self._deadline
                stage, limit, mark = self._stage, self._limit, self._mark
            if not late:
                continue
            print(f"[bench] rank {self.rank}: stage '{stage}' exceeded {limit:.0f} s, leaving", file=sys.stderr, flush=True)
            if not self.have_line:
                os._exit(3)
            if self.rank == 0 and self.fallback is not None:
                line = dict(self.fallback)
                line.update(mark or {"watchdog": f"{stage} timed out after {limit:.0f} s"})
                os.write(self.json_fd, (json.dumps(line) + "\n").encode())
            os._exit(0)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, script=None, need_gpus=True):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, before anything
    in this process touches a GPU.  Rank 0's standard output (the one JSON line) is relayed; the other ranks write to
    standard error.  A rank that fails ends the job: the remaining ranks are stopped by their process ids and the exit
    status is the failed rank's.  (`script`, `need_gpus`: the CPU tests start a stand-in worker over gloo.)"""
    import subprocess
    script = script or os.path.abspath(__file__)
    if need_gpus:
        import torch                                   # device_count() does not initialise the GPU
        have = torch.cuda.device_count()
        if have < n:
            print(f"[bench] --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr, flush=True)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", script] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    line = []
    reader = threading.Thread(target=lambda: line.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    status, deadline = 0, time.monotonic() + 1500
    live = set(range(n))
    while live and status == 0:
        time.sleep(0.2)
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                print(f"[bench] rank {r} ended with status {rc}", file=sys.stderr, flush=True)
                status = rc if rc > 0 else 1
        if time.monotonic() > deadline:
            print("[bench] ranks still running after 1500 s", file=sys.stderr, flush=True)
            status = 4
    for r in live:                                     # only after a failure: stop what is left, by process id
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    reader.join(timeout=10)
    out = (line[0] if line else b"").decode()
    if out.strip():
        sys.stdout.write(out)
        sys.stdout.flush()
    elif status == 0:
        print("[bench] rank 0 printed nothing", file=sys.stderr, flush=True)
        status = 5
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the pixel configurations timed after the headline workload (one GPU only)")
    ap.add_argument("--eager", action="store_true", help="launch kernels one by one instead of replaying a HIP graph")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: 4096 environments and a global batch of 4096 split over the ranks (default: 4096 each per GPU)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks from this process even for --gpus 1 (rehearsal of the launcher-less multi-GPU start)")
    ap.add_argument("--stage-limit", type=float, default=float(os.environ.get("RLGAN_BENCH_STAGE_LIMIT", "90")),
                    help="seconds any one stage (set-up, a timed form, the kernel pass) may take before the watchdog ends the run")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        argv = [a for a in sys.argv[1:] if a != "--spawn"]
        sys.exit(spawn_ranks(args.gpus, argv))

    # Native libraries write to file descriptor 1 as well (RCCL prints a version banner when its first
    # communicator comes up): keep the real stdout aside for the one JSON line and send the rest to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): running with {world}", file=sys.stderr, flush=True)
    dog = Watchdog(rank, json_fd)
    dog.arm(max(args.stage_limit, 300), "set-up")      # the first `import torch` on a fresh box alone can take minutes

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1 or "RANK" in os.environ:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import rl_gan_amd as R
    n_env, batch = (N_ENV // world, BATCH // world) if args.strong else (N_ENV, BATCH)
    if args.strong and n_env * world != N_ENV:
        print("[bench] --strong needs a rank count that divides 4096", file=sys.stderr, flush=True)
        sys.exit(2)
    cfg = R.LearnerConfig(n_env=n_env, batch_size=batch, memory_capacity=CAPACITY, seed=1)
    L = R.CuriousLearner(cfg)
    L.fill()                                            # 257 acting steps: the ring must be full before learning

    # a one-rank job started through a launcher (RANK set) takes the multi-rank code paths below as well, which is how
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

    def measure(steps, soak=0.3, budget=0.25):
        """A short region (--steps 20 is under 4 ms) cannot raise the clocks by itself and one host hiccup is a large
        share of it: besides the --warmup steps the loop first runs for a fixed `soak` seconds, then the region of exactly
        `steps` steps is timed `repeats` times back to back and the MEDIAN region is the one reported."""
        if not multi:
            t_end = time.perf_counter() + soak
            while time.perf_counter() < t_end:
                L.step()
            torch.cuda.synchronize()
        else:
            # every step holds collectives, so all ranks must take the same number of steps: time 50 of them (maximum
            # over ranks, the same value everywhere) and derive the count for `soak` seconds from that
            per_step = timed_region(50) / 50
            for _ in range(min(20000, int(soak / max(per_step, 1e-6)) + 1)):
                L.step()
        repeats = max(3, min(25, int(round(budget / (steps * 2e-4)))) | 1)     # odd; about a quarter of a second in all
        regions = sorted(timed_region(steps) for _ in range(repeats))
        return regions[len(regions) // 2], regions[0], regions[-1], repeats

    def headline(dt, dt_min, dt_max, repeats, launch):
        return {
            "metric": "learner updates/sec (batch 4096)", "value": round((1 if args.strong else world) * args.steps / dt, 2), "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "repeats": repeats, "ms_per_step_min": round(dt_min / args.steps * 1e3, 4), "ms_per_step_max": round(dt_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "curious_gan_ddpg_arm_b4096", "envs_per_gpu": n_env, "batch_per_gpu": batch,
                       "replay_rows_per_gpu": CAPACITY, "global_batch": batch * world, "parallelism": f"dp{world}", "launch": launch},
            "env_steps_per_s": round(world * n_env * args.steps / dt, 1),
        }

    def have_line(line):
        """From here on a hang in a later stage costs that stage only: the watchdog prints `line`."""
        dog.fallback = line
        dog.have_line = True

    # With several ranks the first timing is of kernels launched one by one: it needs nothing of RCCL beyond plain
    # all-reduces, so a line exists before graph capture of collectives (never run with peers before its first multi-GPU
    # box) is attempted.
    for _ in range(args.warmup):
        L.step()
    launch = "eager"
    if multi or args.eager:
        dog.arm(args.stage_limit, "eager timing")
        dt, dt_min, dt_max, repeats = measure(args.steps, soak=0.1 if not args.eager else 0.3, budget=0.05 if not args.eager else 0.25)
        out = headline(dt, dt_min, dt_max, repeats, "eager")
        have_line(out)
    if not args.eager:
        dog.arm(args.stage_limit, "graph capture and timing", mark={"graph_launch": "timed out"})
        try:
            L.capture()                                 # one iteration = one HIP graph replay
            launch = "graph"
        except Exception as e:                          # keep the run alive: same kernels, launched one by one
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
            L._graph = None
            torch.cuda.synchronize()
        if launch == "graph" or not dog.have_line:
            for _ in range(args.warmup):
                L.step()
            dt, dt_min, dt_max, repeats = measure(args.steps)
            eager_ms = out["ms_per_step"] if dog.have_line else None
            out = headline(dt, dt_min, dt_max, repeats, launch)
            if eager_ms is not None:
                out["eager_ms_per_step"] = eager_ms
            have_line(out)

    # per-kernel durations, measured live on the launch stream over a second, eager pass of the same loop
    dog.arm(args.stage_limit, "per-kernel event pass")
    graphs, L._graph = L._graph, None
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
    # median over the eager pass: robust against the first launches (lazy work-array setup)
    avg_ms = {k: sorted(v)[len(v) // 2] for k, v in times.items() if k in MACS_PER_ROW}
    # with more than one rank: the gradient all-reduces of the two phases (name = bucket size), same event timing
    exchange_ms = {k: sorted(v)[len(v) // 2] for k, v in times.items() if k.startswith("exchange_")}
    dom = max(avg_ms, key=avg_ms.get)
    flop = 2.0 * MACS_PER_ROW[dom] * batch
    achieved = flop / (avg_ms[dom] * 1e-3) / 1e12
    traffic, per_iter, busy, source = committed_counters(dom)
    out = dict(out)
    out.update({
        "acting_only_env_steps_per_s": round(acting, 1),
        "kernel_ms": {k: round(v, 4) for k, v in sorted(avg_ms.items())},
        "losses": {k: round(v, 6) for k, v in losses.items()},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "algorithmic_gflop": round(flop / 1e9, 4),
                     "timing": "HIP events on the launch stream around each launch of an eager pass of the same loop, median",
                     "traffic": traffic, "mfma_busy": busy, "traffic_per_iteration": per_iter, "counters_from": source},
    })
    if exchange_ms:
        out["exchange_ms"] = {k: round(v, 4) for k, v in sorted(exchange_ms.items())}
    have_line(out)

    if L.engine.exchange.enabled and launch == "graph":
        # the other form of the multi-rank update (two chains on two streams, one exchange per step), same run, same
        # measurement: the scaling record then shows which form the exchange latency favours.  It has never run with
        # peers, so it is timed last and under the watchdog: if it hangs, the line above is printed without it.
        dog.arm(args.stage_limit, "two-chain form", mark={"update_forms": "two_chains timed out"})
        try:
            L.engine.set_chains(2)
            L._graph = None
            L.capture()
            dt2, dt2_min, dt2_max, rep2 = measure(args.steps)
            forms = {"fused_phases_ms_per_step": out["ms_per_step"], "two_chains_ms_per_step": round(dt2 / args.steps * 1e3, 4)}
            if dt2 < 0.99 * out["ms_per_step"] * 1e-3 * args.steps or os.environ.get("RLGAN_BENCH_REPORT_FORM") == "two_chains":   # (the variable: rehearsal of this branch on one GPU)
                # both forms give the same bits (tests/test_gpu_dp.py); the line reports the one this machine's exchange
                # latency favours, and says which
                better = headline(dt2, dt2_min, dt2_max, rep2, launch)
                better["config"]["update_form"] = "two_chains"
                out.update({k: better[k] for k in ("value", "ms_per_step", "repeats", "ms_per_step_min", "ms_per_step_max",
                                                   "env_steps_per_s", "config")})
            out["update_forms"] = forms
        except Exception as e:
            print(f"[bench] two-chain form not timed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            out["update_forms"] = f"two_chains failed: {type(e).__name__}"
            torch.cuda.synchronize()
        L.engine.set_chains(1)
        have_line(out)
    dog.disarm()

    if rank == 0:
        if world == 1 and not multi and not args.no_other_workloads:
            # the reference's pixel configurations, a couple of seconds each (tools/workloads.py); reported beside
            # the headline, never part of `value`
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from workloads import run_all
            del L, graphs
            torch.cuda.empty_cache()
            out["other_workloads"] = run_all(BATCH)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dog.arm(30, "process-group shutdown", mark={})
        if rank == 0:
            dog.fallback = None                        # the line is out: a late expiry must not print it twice
        dist.destroy_process_group()
        dog.disarm()


if __name__ == "__main__":
    main()
