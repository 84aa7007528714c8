#!/usr/bin/env python3
"""Time of the pixel generator's and discriminator's training steps (Conv_model.py:270-281,533-541) at batch B,
states read from the uint8 replay ring in place."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rl_gan_amd as R
from kbench import timeit

for B in [int(v) for v in sys.argv[1:]] or [32, 4096]:
    rng = np.random.RandomState(0)
    a_dim = 6
    st = R.ConvEncoder(42, 42, 4, R.universe_head(2), rng=rng, dense=256, bias_init=0.001)
    ob = R.single_frame_encoder(42, 2, rng=rng)
    G = R.PixelGenerator(256, a_dim, 1e-4, ob, rng=rng)
    D = R.PixelDiscriminator(a_dim, 1e-4, st, ob, rng=rng)
    mem = R.FrameMemory(max(5000, B), (42, 42, 4), rng=R.MT19937Device(1))
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for _ in range((mem.capacity + 1 + B - 1) // B):
        f = torch.randint(0, 256, (min(B, mem.capacity),) + (42, 42, 4), dtype=torch.uint8, device="cuda", generator=g)
        mem.store_transitions(f, f)
    idx = mem.sample_indices(B)
    a = torch.rand(B, a_dim, device="cuda") * 2 - 1
    s_next1 = mem.gather(idx, "s_")[..., 3:].contiguous()          # last frame of s' (Conv_model.py:536)
    phis = st.forward(mem.ref(idx)).clone()
    img = G.predict_batch(phis, a).clone()
    tg = timeit(lambda: G.learn(phis, a, s_next1))
    td = timeit(lambda: D.learn(img, s_next1, mem.ref(idx), a))
    tp = timeit(lambda: G.predict_batch(phis, a))
    tr = timeit(lambda: D.determine_batch(mem.ref(idx), a, img))
    actions = R.DeviceMemory(mem.capacity + 1, a_dim)
    for i in range(0, mem.capacity + 1, 4096):
        k = min(4096, mem.capacity + 1 - i)
        actions.store_rows(torch.rand(k, a_dim, device="cuda") * 2 - 1)
    tr_ = R.PixelGanTrainer(mem, actions, G, D, B)
    tr_.sample()
    te = timeit(tr_.step)
    tr_.capture()
    tgph = timeit(lambda: (tr_.sample(), tr_.step()))
    print(f"B={B:5d}: GAN iteration (index draw, G step, prediction, D step): one by one {te:9.1f} us, as a graph {tgph:9.1f} us"
          f" = {B / tgph * 1e6:9.0f} transitions/s")
    print(f"B={B:5d}: generator.learn {tg:9.1f} us   discriminator.learn {td:9.1f} us   predict_batch {tp:8.1f} us   determine_batch {tr:8.1f} us"
          f"   ({B / (tg + td) * 1e6:9.0f} transitions/s through both steps)")
    # the rest of the reference's learning iteration (Curious_GAN_mario.py:96-98): inverse model, critic, actor
    A_pred = R.StateActionPredictor(a_dim, 1e-4, st, rng=rng)
    actor = R.PixelActor(a_dim, 1e-4, 1100, st, rng=rng)
    critic = R.PixelCritic(a_dim, 1e-4, 0.99, 1000, actor, st, rng=rng)
    s, s_next = mem.gather(idx, "s"), mem.gather(idx, "s_")
    r = torch.rand(B, 1, device="cuda")
    phis_next = st.forward(s_next).clone()
    ta = timeit(lambda: A_pred.learn(s, s_next, a))
    tc = timeit(lambda: critic.learn(None, a, r, None, phis=phis, phis_next=phis_next))
    tac = timeit(lambda: actor.learn(None, critic, phis=phis))
    scal = R.DeviceMemory(mem.capacity + 1, a_dim + 2)
    for i in range(0, mem.capacity + 1, 4096):
        k = min(4096, mem.capacity + 1 - i)
        scal.store_rows(torch.rand(k, a_dim + 2, device="cuda") * 2 - 1)
    full = R.PixelGanTrainer(mem, scal, G, D, B, a_pred=A_pred, actor=actor, critic=critic)
    full.sample()
    tfe = timeit(full.step)
    full.capture()
    tfg = timeit(lambda: (full.sample(), full.step()))
    print(f"B={B:5d}: A_pred.learn {ta:9.1f} us   critic.learn (features given) {tc:8.1f} us   actor.learn (features given) {tac:8.1f} us")
    print(f"B={B:5d}: whole learning iteration (GAN half + inverse model + critic + actor): one by one {tfe:9.1f} us, as a graph {tfg:9.1f} us"
          f" = {B / tfg * 1e6:9.0f} transitions/s")
    # the acting side of the loop for B environments at once (Curious_GAN_mario.py:63-69): action, predicted frame,
    # intrinsic reward, LSTM state; the frames come from the uint8 ring
    c0 = torch.zeros(B, 256, device="cuda"); h0 = torch.zeros(B, 256, device="cuda")
    def acting():
        ph = st.forward(mem.ref(idx))
        act = actor.actions(ph)[:, :a_dim].contiguous()
        g_img = G.predict_batch(ph, act)
        D.determine_batch(mem.ref(idx), act, g_img)
        actor.LSTM_unit.step_batch(ph, c0, h0)
    tact = timeit(acting)
    print(f"B={B:5d}: acting side (encoder, actor, predicted frame, 1 - D, LSTM step) {tact:9.1f} us = {B / tact * 1e6:9.0f} env-steps/s of compute")
