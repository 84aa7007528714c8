"""Replay ring oracle against the reference's Memory classes (tests/golden/memory_ring.npz)."""
import os

import numpy as np

from oracle.memory import Memory

from conftest import GOLDEN


def test_ring_layout_and_sample():
    d = np.load(os.path.join(GOLDEN, "memory_ring.npz"))
    cap = int(d["capacity"])
    for tag, dims in (("4", 17), ("6", 25)):
        m = Memory(cap, dims, rng=np.random.RandomState(int(d["sample_seed" + tag])))
        for row in d["rows" + tag]:
            m.store_row(row)
        assert m.pointer == int(d["pointer" + tag])
        np.testing.assert_array_equal(m.data, d["data" + tag])
        batch, idx = m.sample(16)
        np.testing.assert_array_equal(batch, d["batch" + tag])
        assert idx.dtype == np.int64


def test_field_slicing_matches_reference_columns():
    # Curious_GAN.py:200-205 on the 25-column layout
    row = np.arange(25.0)[None]
    S, A = 7, 2
    assert row[:, :S].tolist()[0] == list(range(0, 7))
    assert row[:, S:2 * S].tolist()[0] == list(range(7, 14))
    assert row[:, 2 * S:3 * S].tolist()[0] == list(range(14, 21))
    assert row[:, -2 - A:-2].tolist()[0] == [21, 22]
    assert row[:, -2:-1].tolist()[0] == [23] and row[:, -1:].tolist()[0] == [24]
