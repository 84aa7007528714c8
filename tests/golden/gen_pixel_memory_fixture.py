#!/usr/bin/env python3
"""Generates tests/golden/pixel_memory.npz by running the reference's own pixel replay
(`Conv_model.Memory`, Conv_model.py:102-157) in this container, fed with frame stacks scaled the way
`env_wrapper.BufferedObsEnv` hands them out (uint8 * 1/255 as float32, env_wrapper.py:44,58).

The reference is imported from /root/reference/GAN-RL with an inert stand-in for the absent
`tensorflow` module name (Conv_model.py:1 imports it; `Memory` never touches it).  Only the resulting
arrays are committed; no reference source leaves that directory.

    python tests/golden/gen_pixel_memory_fixture.py
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
    cap, n, shape = 7, 19, (6, 6, 2)
    rng = np.random.RandomState(3)
    frames_s = rng.randint(0, 256, (n,) + shape).astype(np.uint8)
    frames_s_ = rng.randint(0, 256, (n,) + shape).astype(np.uint8)
    scale = 1.0 / 255                                    # env_wrapper.py:44
    M = Conv_model.Memory(cap)
    pointers, lens = [], []
    for i in range(n):
        s = frames_s[i].astype(np.float32) * scale       # env_wrapper.py:58
        s_ = frames_s_[i].astype(np.float32) * scale
        M.store_transition(s, 0.0, np.zeros((1, 1)), np.zeros(2), 0.0, 0.0, s_, np.zeros((1, 1)), np.zeros(2))
        pointers.append(M.pointer)
        lens.append(M.len)
    np.random.seed(21)
    out = M.sample(12)
    np.random.seed(21)
    idx = np.random.choice(M.len, size=12)               # the draw inside sample(), Conv_model.py:147
    np.savez(os.path.join(HERE, "pixel_memory.npz"), capacity=cap, frames_s=frames_s, frames_s_=frames_s_,
             data_s=np.array(M.data_s), data_s_=np.array(M.data_s_), pointers=np.array(pointers), lens=np.array(lens),
             sample_seed=21, indices=idx, b_s=out[0], b_s_=out[6])
    print("written", os.path.join(HERE, "pixel_memory.npz"))


if __name__ == "__main__":
    main()
