#!/usr/bin/env python3
"""Probe (diagnostic, not part of the product): PyTorch's symmetric-memory one-shot all-reduce between the ranks of a
torchrun job that share the one GPU of a test box (gloo process group; the buffers are exchanged as IPC handles):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29540 tools/symm_probe.py"""
import os, time, torch, torch.distributed as dist
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dist.init_process_group("gloo" if world > 1 else "nccl")
import torch.distributed._symmetric_memory as sm
name = dist.group.WORLD.group_name
n = 134 * 1024
t = sm.empty(n, dtype=torch.float32, device="cuda")
h = sm.rendezvous(t, name)
print(rank, "rendezvous ok, world", h.world_size, flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(rank)
for it in range(3):
    t.copy_(torch.randn(n, device="cuda", generator=g))
    mine = t.clone().cpu()
    out = torch.ops.symm_mem.one_shot_all_reduce(t, "sum", name)
    torch.cuda.synchronize()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(gathered[1:], gathered[0].clone())
    print(rank, "iteration", it, "max |err|", float((out.cpu() - want).abs().max()), "bit-equal to the rank-ordered sum:", bool(torch.equal(out.cpu(), want)), flush=True)
dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    out = torch.ops.symm_mem.one_shot_all_reduce(t, "sum", name)
torch.cuda.synchronize()
print(rank, f"one_shot_all_reduce of {n * 4 // 1024} KiB: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call (eager)", flush=True)
dist.destroy_process_group()
