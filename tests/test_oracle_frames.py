"""The frame-stack replay restatement (oracle/frames.py) against the arrays the reference's own
`Conv_model.Memory` produced (tests/golden/pixel_memory.npz; Conv_model.py:102-157, env_wrapper.py:44,58)."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import frames as OF


def _replay(d):
    m = OF.FrameMemory(int(d["capacity"]))
    pointers, lens = [], []
    for s, s_ in zip(d["frames_s"], d["frames_s_"]):
        m.store_transition(s, s_)
        pointers.append(m.pointer)
        lens.append(m.len)
    return m, pointers, lens


def test_slots_pointer_and_len_follow_the_reference():
    d = np.load(os.path.join(GOLDEN, "pixel_memory.npz"))
    m, pointers, lens = _replay(d)
    np.testing.assert_array_equal(pointers, d["pointers"])
    np.testing.assert_array_equal(lens, d["lens"])
    assert np.array(m.data_s).dtype == np.float32
    np.testing.assert_array_equal(np.array(m.data_s), d["data_s"])
    np.testing.assert_array_equal(np.array(m.data_s_), d["data_s_"])
    # capacity + 1 slots; the last one keeps the transition stored at pointer == capacity for ever
    cap = int(d["capacity"])
    assert len(m.data_s) == cap + 1
    np.testing.assert_array_equal(m.data_s[cap], OF.scaled(d["frames_s"][cap]))


def test_sample_draws_the_reference_indices_and_frames():
    d = np.load(os.path.join(GOLDEN, "pixel_memory.npz"))
    m, _, _ = _replay(d)
    np.random.seed(int(d["sample_seed"]))
    idx, b_s, b_s_ = m.sample(len(d["indices"]))
    np.testing.assert_array_equal(idx, d["indices"])
    np.testing.assert_array_equal(b_s, d["b_s"])
    np.testing.assert_array_equal(b_s_, d["b_s_"])


def test_slot_rule():
    m = OF.FrameMemory(5)
    assert [m.slot_of(p) for p in range(13)] == [0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 0, 1, 2]
