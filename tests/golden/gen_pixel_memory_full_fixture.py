#!/usr/bin/env python3
"""Generates tests/golden/pixel_memory_full.npz by running the reference's own pixel replay
(`Conv_model.Memory`, Conv_model.py:102-157) in this container with all nine fields of a transition filled the way
the training loop fills them (Curious_GAN_mario.py:63-75): frame stacks as uint8 * 1/255 float32
(env_wrapper.py:44,58), the generated frame, the LSTM states [2, 1, units], the one-hot action, the rewards and the
dense action.

The reference is imported from /root/reference/GAN-RL with an inert stand-in for the absent `tensorflow` module
name (Conv_model.py:1 imports it; `Memory` never touches it).  Only the resulting arrays are committed.

    python tests/golden/gen_pixel_memory_full_fixture.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/GAN-RL"


def main():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    sys.path.insert(0, REF)
    import Conv_model
    cap, n, shape, units, ac = 6, 17, (6, 6, 4), 8, 3
    rng = np.random.RandomState(5)
    f32 = lambda v: np.asarray(v, dtype=np.float32)
    frames_s = rng.randint(0, 256, (n,) + shape).astype(np.uint8)
    frames_s_ = rng.randint(0, 256, (n,) + shape).astype(np.uint8)
    g = f32(rng.uniform(-0.2, 1.1, (n,) + shape[:2] + (1,)))
    lstm_s, lstm_s_ = f32(rng.standard_normal((n, 2, 1, units))), f32(rng.standard_normal((n, 2, 1, units)))
    a = np.eye(ac, dtype=np.int64)[rng.randint(0, ac, n)]
    r = f32(rng.randint(0, 3, n))
    curious_r = f32(rng.uniform(0, 1, (n, 1)))
    dense_a = rng.randint(0, 14, n)
    M = Conv_model.Memory(cap)
    for i in range(n):
        s = frames_s[i].astype(np.float32) * (1.0 / 255)
        s_ = frames_s_[i].astype(np.float32) * (1.0 / 255)
        M.store_transition(s, g[i], [lstm_s[i, 0], lstm_s[i, 1]], a[i], float(r[i]), curious_r[i], s_,
                           [lstm_s_[i, 0], lstm_s_[i, 1]], int(dense_a[i]))
    np.random.seed(9)
    out = M.sample(10)
    names = ("b_s", "b_g", "b_lstm_s", "b_a", "b_r", "b_curious_r", "b_s_", "b_lstm_s_again", "b_d_a")
    np.savez(os.path.join(HERE, "pixel_memory_full.npz"), capacity=cap, units=units, ac=ac, frames_s=frames_s, frames_s_=frames_s_,
             g=g, lstm_s=lstm_s, lstm_s_=lstm_s_, a=a, r=r, curious_r=curious_r, dense_a=dense_a, sample_seed=9, pointer=M.pointer,
             len=M.len, **{k: np.asarray(v) for k, v in zip(names, out)})
    print("written", {k: np.asarray(v).shape for k, v in zip(names, out)})


if __name__ == "__main__":
    main()
