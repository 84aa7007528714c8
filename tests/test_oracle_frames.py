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


def test_all_nine_fields_against_the_reference_made_fixture():
    """oracle.frames.PixelMemory against what the reference's own `Conv_model.Memory` returned for the same stores
    and the same NumPy seed (tests/golden/pixel_memory_full.npz)."""
    import os
    from oracle.frames import PixelMemory
    F = np.load(os.path.join(os.path.dirname(__file__), "golden", "pixel_memory_full.npz"))
    M = PixelMemory(int(F["capacity"]))
    for i in range(F["frames_s"].shape[0]):
        M.store_full(F["frames_s"][i], F["g"][i], [F["lstm_s"][i, 0], F["lstm_s"][i, 1]], F["a"][i], float(F["r"][i]), F["curious_r"][i],
                     F["frames_s_"][i], [F["lstm_s_"][i, 0], F["lstm_s_"][i, 1]], int(F["dense_a"][i]))
    assert M.pointer == int(F["pointer"]) and M.len == int(F["len"])
    out = M.sample_full(10, np.random.RandomState(int(F["sample_seed"])))
    names = ("b_s", "b_g", "b_lstm_s", "b_a", "b_r", "b_curious_r", "b_s_", "b_lstm_s_again", "b_d_a")
    for k, v in zip(names, out):
        np.testing.assert_array_equal(np.asarray(v), F[k], err_msg=k)
