#!/usr/bin/env python3
"""The pixel Curious-GAN iteration replayed many times: every loss and every parameter finite at the end, and the time per
iteration at the start and at the end.   python tools/soak_pixel.py [iterations] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import workloads

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
step = workloads.pixel_curious_gan(batch)
tr = step.trainer


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


first = timed(100)
done = 100
while done < iters - 100:
    n = min(500, iters - 100 - done)
    timed(n)
    done += n
    print(f"{done} iterations", flush=True)
last = timed(100)


NAMES = ("canon", "grad", "wmat", "wfrag", "wdx", "loss_out", "losses", "gamma", "beta", "m", "v")


def tensors(obj, seen, name="", depth=0):
    """(attribute name, tensor) of every float tensor under obj whose attribute is one of NAMES (parameters, gradients,
    optimizer moments, losses: buffers that are fully written every iteration; scratch space is not)."""
    if id(obj) in seen or depth > 8:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_floating_point() and name in NAMES:
            yield name, obj
        return
    if isinstance(obj, (list, tuple)):
        for o in obj:
            yield from tensors(o, seen, name, depth + 1)
    elif isinstance(obj, dict):
        for k, o in obj.items():
            yield from tensors(o, seen, k if isinstance(k, str) else name, depth + 1)
    elif hasattr(obj, "__dict__") and not isinstance(obj, type):
        for k, o in vars(obj).items():
            yield from tensors(o, seen, k.lstrip("_"), depth + 1)


seen = set()
n_t = n_bad = 0
for owner in (tr.G, tr.D, tr.a_pred, tr.actor, tr.critic):
    for name, x in tensors(owner, seen):
        if x.numel():
            n_t += 1
            if not bool(torch.isfinite(x).all()):
                n_bad += 1
                print("not finite:", type(owner).__name__, name, tuple(x.shape))
print(f"{iters} iterations at batch {batch}: {first:.3f} ms per iteration at the start, {last:.3f} at the end; "
      f"{n_t} float tensors checked, {n_bad} with a non-finite entry")
sys.exit(1 if n_bad else 0)
