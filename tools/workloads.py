#!/usr/bin/env python3
"""The two pixel configurations of the reference as timed workloads at batch 4096, synthetic frames, random weights:

* `conv_ddpg_doom_84_b4096` -- Conv_DDPG.py:82-87 on Conv_model.py:15-24,569-784: 84x84x4 float32 frames -> the
  four-convolution "doom" head -> 64 features -> the 256-unit ELU layer -> critic.learn then actor.learn, replayed as
  one HIP graph.
* `pixel_curious_gan_universe_42_b4096` -- the learning iteration of Curious_GAN_mario.py:85-98 on 42x42x4 uint8
  frame stacks: index draw, generator step (decoder included), prediction, discriminator step, inverse model,
  critic and actor steps, replayed as one HIP graph.

* `conv_universe_84_b4096` -- the same update behind the two-convolution "universe" head on 84x84x4 float32 frames
  (Conv_model.py:36-56): 41x41x32 -> 20x20x32 = 12,800 features -> the 256-unit ELU layer.
* `pixel_curious_gan_universe_42_b32` -- the pixel iteration at the reference's own batch of 32 (Curious_GAN_mario.py:393).
* `conv_universe_84_b32` -- the 84x84 universe-head update at that batch of 32.

bench.py times them after its headline workload (`other_workloads` in its JSON line); run directly it prints them.
Every entry carries a `roofline` object: the algorithmic multiply-accumulates of one step (`step_macs`, below) against the
dense fp32 matrix peak, and, for the workload whose frames cost more time at the HBM peak than its products at the matrix
peak, the frame bytes against the HBM peak."""
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


PEAK_FP32_TFLOPS = 157.3     # dense fp32 matrix peak of one MI355X
PEAK_HBM_GBS = 8000.0


def conv_macs(h, w, c, head):
    """Forward multiply-accumulates per frame of a valid-padding convolution stack [(filters, kernel, stride), ...]:
    per layer and in total, plus the first layer's alone (its input gradient is never needed) and the output shape."""
    per = []
    for f, k, s in head:
        h, w = (h - k) // s + 1, (w - k) // s + 1
        per.append(h * w * f * k * k * c)
        c = f
    return per, (h, w, c)


def mlp_macs(widths):
    return sum(a * b for a, b in zip(widths[:-1], widths[1:]))


def ddpg_step_macs(feat, a_dim):
    """critic.learn + actor.learn on `feat` encoder features (Conv_model.py:610-663,711-784): pixel critic
    [feat + a] -> 200 -> 200 -> 10 -> 1, pixel actor feat -> 200 -> 200 -> 10 -> a.  Critic step: a'(phi'), q'(phi', a'),
    q(phi, a) forward, then input + weight gradients of q (2 x forward).  Actor step: a(phi), q(phi, a) forward, dq/da
    (1 x the critic), input + weight gradients of the actor (2 x forward)."""
    C, A = mlp_macs([feat + a_dim, 200, 200, 10, 1]), mlp_macs([feat, 200, 200, 10, a_dim])
    return (A + 2 * C + 2 * C) + (A + C + C + 2 * A)


def encoder_update_macs(hw, head, a_dim=6, feat=256):
    """One step of the Conv_DDPG-shaped workloads per minibatch row: the encoder on s and on s' (forward only: neither
    step trains it), then the two steps on its features."""
    per, (h, w, c) = conv_macs(hw, hw, 4, head)
    enc = sum(per) + h * w * c * feat
    return 2 * enc + ddpg_step_macs(feat, a_dim)


def pixel_iteration_macs(a_dim=6):
    """One learning iteration of the pixel Curious-GAN on 42x42 frames per minibatch row, as PixelGanTrainer runs it
    (Curious_GAN_mario.py:85-98).  Backward = input gradient (1 x forward) + weight gradient (1 x forward); a first
    convolution layer has no input gradient unless something trainable or wanted sits in front of it."""
    st_per, (h, w, c) = conv_macs(42, 42, 4, [(32, 3, 2)] * 2)
    st_head = h * w * c * 256
    E_st = sum(st_per) + st_head                                   # state encoder, 1,870,848
    ob_per, _ = conv_macs(42, 42, 1, [(32, 3, 2)] * 2)
    E_ob = sum(ob_per) + st_head                                   # single-frame encoder
    dec = 200 * 128 * 9 + 9 * 128 * 64 * 16 + 64 * 64 * 32 * 25 + 361 * 32 * 36   # decoder 1x1x200 -> 42x42x1 (Conv_model.py:457-473)
    G_l1 = (256 + a_dim) * 200
    G_fwd = G_l1 + dec
    D_mlp = mlp_macs([256 + 256 + a_dim, 200, 10, 1])
    inv = mlp_macs([512, 500, 100, a_dim])                         # StateActionPredictor (Conv_model.py:787-880)
    st_bwd, ob_bwd = 2 * E_st - st_per[0], 2 * E_ob - ob_per[0]    # trained encoders: no input gradient of layer 1
    total = E_st + E_ob                                            # features of s; features of the real last frame
    total += G_fwd + E_ob + E_ob + 2 * dec + G_l1                  # G.learn: predict, encode, dX through the encoder, decoder backward, l1 dW
    total += G_fwd + E_ob                                          # prediction after the step, encoded for D
    total += 2 * 3 * D_mlp + 2 * ob_bwd + st_bwd                   # D.learn: real + fake rows; both encoders trained
    total += 2 * E_st + 3 * inv + 2 * st_bwd                       # inverse model on (s, s'), trains the state encoder
    total += 2 * E_st + ddpg_step_macs(256, a_dim)                 # critic.learn, actor.learn on fresh features of s, s'
    return total


def roofline(macs_per_row, B, ms, frame_bytes_per_row=0):
    """frac of the bound that costs the step more time at its peak: the matrix products, or the bytes of the frames."""
    flop = 2.0 * macs_per_row * B
    t_mfma = flop / (PEAK_FP32_TFLOPS * 1e12)
    t_hbm = frame_bytes_per_row * B / (PEAK_HBM_GBS * 1e9)
    out = {"algorithmic_gflop": round(flop / 1e9, 3), "mfma_tflops": round(flop / (ms * 1e-3) / 1e12, 2),
           "mfma_frac": round(t_mfma / (ms * 1e-3), 4)}
    if frame_bytes_per_row:
        out.update({"algorithmic_mbytes": round(frame_bytes_per_row * B / 1e6, 1),
                    "hbm_gbs": round(frame_bytes_per_row * B / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(t_hbm / (ms * 1e-3), 4)})
    if t_hbm > t_mfma:
        out.update({"bound": "hbm", "achieved": out["hbm_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": out["hbm_frac"]})
    else:
        out.update({"bound": "mfma", "achieved": out["mfma_tflops"], "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": out["mfma_frac"]})
    return out


def conv_ddpg_doom(B=4096, a_dim=6, head=None):
    import numpy as np
    import torch
    import rl_gan_amd as R
    rng = np.random.RandomState(0)
    enc = R.ConvEncoder(84, 84, 4, head or R.DOOM_HEAD, rng=rng, dense=256, bias_init=0.001)
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
        phis_n = enc.forward(s_n)                  # every pass returns buffers of its own: no copy needed to keep these
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
    step.trainer = tr                                       # for tools/soak_pixel.py
    return step


def conv_universe_84(B=4096, a_dim=6):
    import rl_gan_amd as R
    return conv_ddpg_doom(B, a_dim, head=R.universe_head(2))


def doom_head():
    return [(8, 5, 4), (16, 3, 2), (32, 3, 2), (64, 3, 2)]       # Conv_model.py:15-22


FRAME_BYTES_84 = 2 * 84 * 84 * 4 * 4          # s and s' as float32 frame stacks, read once each per step

# name, builder, batch (None: the caller's), multiply-accumulates per minibatch row, frame bytes per row
WORKLOADS = (
    ("conv_ddpg_doom_84_b4096", conv_ddpg_doom, None, lambda: encoder_update_macs(84, doom_head()), FRAME_BYTES_84),
    ("conv_universe_84_b4096", conv_universe_84, None, lambda: encoder_update_macs(84, [(32, 3, 2)] * 2), FRAME_BYTES_84),
    ("pixel_curious_gan_universe_42_b4096", pixel_curious_gan, None, pixel_iteration_macs, 0),
    ("pixel_curious_gan_universe_42_b32", pixel_curious_gan, 32, pixel_iteration_macs, 0),
    ("conv_universe_84_b32", conv_universe_84, 32, lambda: encoder_update_macs(84, [(32, 3, 2)] * 2), FRAME_BYTES_84),
)


def run_all(B=4096, seconds=1.5, only=None):
    """[{workload, value (updates/s at its batch), ms_per_step, transitions_per_s, steps, roofline}] -- a failed workload
    is reported with its error instead of a value, never silently dropped."""
    import gc
    import torch
    out = []
    for name, build, fixed_b, macs, frame_bytes in WORKLOADS:
        if only and name != only:
            continue
        b = fixed_b or B
        try:
            step = build(b)
            ms, n = time_steps(step, seconds if fixed_b is None else seconds / 3)
            out.append({"workload": name, "value": round(1e3 / ms, 2), "unit": "updates/s", "ms_per_step": round(ms, 4), "batch": b,
                        "transitions_per_s": round(b * 1e3 / ms, 1), "steps": n, "dtype": "f32", "data": "synthetic",
                        "roofline": roofline(macs(), b, ms, frame_bytes)})
            del step
        except Exception as e:                                       # noqa: BLE001 -- the headline line must still be printed
            out.append({"workload": name, "error": f"{type(e).__name__}: {e}"})
        gc.collect()
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    print(json.dumps(run_all(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, only=sys.argv[2] if len(sys.argv) > 2 else None), indent=1))
