#!/usr/bin/env python3
"""The two pixel configurations of the reference as timed workloads at batch 4096, synthetic frames, random weights:

* `conv_ddpg_doom_84_b4096` -- Conv_DDPG.py:82-87 on Conv_model.py:15-24,569-784: 84x84x4 float32 frames -> the
  four-convolution "doom" head -> 64 features -> the 256-unit ELU layer -> critic.learn then actor.learn, replayed as
  one HIP graph.
* `pixel_curious_gan_universe_42_b4096` -- the learning iteration of Curious_GAN_mario.py:85-98 on 42x42x4 uint8
  frame stacks: index draw, generator step (decoder included), prediction, discriminator step, inverse model,
  critic and actor steps, replayed as one HIP graph.

bench.py times both after its headline workload (`other_workloads` in its JSON line); run directly it prints them."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_steps(step, seconds=1.5, warm=3):
    """Median of the per-step time over back-to-back regions that together last about `seconds`; milliseconds."""
    import torch
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-5)
    per = max(1, min(50, int(seconds / 5 / one)))
    regions = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per):
            step()
        torch.cuda.synchronize()
        regions.append((time.perf_counter() - t0) / per)
    return sorted(regions)[2] * 1e3, per * 5


def conv_ddpg_doom(B=4096, a_dim=6):
    import numpy as np
    import torch
    import rl_gan_amd as R
    rng = np.random.RandomState(0)
    enc = R.ConvEncoder(84, 84, 4, R.DOOM_HEAD, rng=rng, dense=256, bias_init=0.001)
    actor = R.PixelActor(a_dim, 1e-4, 1100, enc, rng=rng)
    critic = R.PixelCritic(a_dim, 1e-4, 0.99, 1000, actor, enc, rng=rng)
    g = torch.Generator(device="cuda").manual_seed(0)
    s = torch.rand((B, 84, 84, 4), generator=g, device="cuda", dtype=torch.float32)
    s_n = torch.rand((B, 84, 84, 4), generator=g, device="cuda", dtype=torch.float32)
    a = torch.nn.functional.one_hot(torch.randint(0, a_dim, (B,), generator=g, device="cuda"), a_dim).float()
    r = torch.rand((B, 1), generator=g, device="cuda", dtype=torch.float32)

    def body():
        # neither step trains the encoder (Conv_model.py:616,784), so the features of s serve both: two encoder passes per
        # update where the reference's two sess.run calls make three
        phis_n = enc.forward(s_n).clone()
        phis = enc.forward(s)
        critic.learn(None, a, r, None, phis=phis, phis_next=phis_n)
        actor.learn(None, critic, phis=phis)

    # the update is ~90 short launches: replayed as one HIP graph like the pixel Curious-GAN iteration (every counter and
    # step size the kernels read lives on the device, so the replay is the same computation)
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()
    return graph.replay


def pixel_curious_gan(B=4096, a_dim=6):
    import numpy as np
    import torch
    import rl_gan_amd as R
    rng = np.random.RandomState(0)
    st = R.ConvEncoder(42, 42, 4, R.universe_head(2), rng=rng, dense=256, bias_init=0.001)
    ob = R.single_frame_encoder(42, 2, rng=rng)
    G = R.PixelGenerator(256, a_dim, 1e-4, ob, rng=rng)
    D = R.PixelDiscriminator(a_dim, 1e-4, st, ob, rng=rng)
    A_pred = R.StateActionPredictor(a_dim, 1e-4, st, rng=rng)
    actor = R.PixelActor(a_dim, 1e-4, 1100, st, rng=rng)
    critic = R.PixelCritic(a_dim, 1e-4, 0.99, 1000, actor, st, rng=rng)
    mem = R.FrameMemory(max(5000, B), (42, 42, 4), rng=R.MT19937Device(1))
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range((mem.capacity + 1 + B - 1) // B):
        f = torch.randint(0, 256, (min(B, mem.capacity), 42, 42, 4), dtype=torch.uint8, device="cuda", generator=g)
        mem.store_transitions(f, f)
    scal = R.DeviceMemory(mem.capacity + 1, a_dim + 2)
    for i in range(0, mem.capacity + 1, 4096):
        k = min(4096, mem.capacity + 1 - i)
        scal.store_rows(torch.rand(k, a_dim + 2, device="cuda") * 2 - 1)
    tr = R.PixelGanTrainer(mem, scal, G, D, B, a_pred=A_pred, actor=actor, critic=critic)
    tr.sample()
    tr.step()
    tr.capture()

    def step():
        tr.sample()
        tr.step()
    return step


WORKLOADS = (("conv_ddpg_doom_84_b4096", conv_ddpg_doom), ("pixel_curious_gan_universe_42_b4096", pixel_curious_gan))


def run_all(B=4096, seconds=1.5, only=None):
    """[{workload, value (updates/s at batch B), ms_per_step, transitions_per_s, steps}] -- a failed workload is
    reported with its error instead of a value, never silently dropped."""
    import gc
    import torch
    out = []
    for name, build in WORKLOADS:
        if only and name != only:
            continue
        try:
            step = build(B)
            ms, n = time_steps(step, seconds)
            out.append({"workload": name, "value": round(1e3 / ms, 2), "unit": "updates/s", "ms_per_step": round(ms, 4),
                        "transitions_per_s": round(B * 1e3 / ms, 1), "steps": n, "dtype": "f32", "data": "synthetic"})
            del step
        except Exception as e:                                       # noqa: BLE001 -- the headline line must still be printed
            out.append({"workload": name, "error": f"{type(e).__name__}: {e}"})
        gc.collect()
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    print(json.dumps(run_all(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, only=sys.argv[2] if len(sys.argv) > 2 else None), indent=1))
