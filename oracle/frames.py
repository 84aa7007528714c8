"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the pixel variants' replay of frame stacks.

Restates `Conv_model.Memory` (Conv_model.py:102-157) for its image fields and the scaling of
`env_wrapper.BufferedObsEnv` (env_wrapper.py:44,58).  Pinned by tests/golden/pixel_memory.npz, which the
reference's own class produced in the build container (tests/golden/gen_pixel_memory_fixture.py)."""
import numpy as np

SCALE = 1.0 / 255                       # env_wrapper.py:44


def scaled(frames_u8: np.ndarray) -> np.ndarray:
    """`obsNew.astype(np.float32) * self.scale` (env_wrapper.py:58): float32 times a Python float stays float32."""
    return frames_u8.astype(np.float32) * SCALE


class FrameMemory:
    """Slots are appended while `pointer <= capacity` (so there are capacity + 1 of them) and afterwards
    overwritten at `pointer % capacity`; slot `capacity` is therefore never replaced (Conv_model.py:119-143).
    `sample` draws `np.random.choice(len, n)` with len frozen at capacity + 1 (Conv_model.py:143,147)."""

    def __init__(self, capacity: int):
        self.capacity = capacity
        self.data_s, self.data_s_ = [], []
        self.pointer = 0
        self.len = 0

    def slot_of(self, pointer: int) -> int:
        return pointer % self.capacity if pointer > self.capacity else pointer

    def store_transition(self, s_u8: np.ndarray, s_next_u8: np.ndarray) -> None:
        s, s_ = scaled(s_u8), scaled(s_next_u8)
        if self.pointer > self.capacity:                               # Conv_model.py:119-131
            i = self.pointer % self.capacity
            self.data_s[i], self.data_s_[i] = s, s_
            self.pointer += 1
        else:                                                          # Conv_model.py:132-143
            self.data_s.append(s)
            self.data_s_.append(s_)
            self.pointer += 1
            self.len = self.pointer

    def sample(self, n: int, rs=np.random):
        idx = rs.choice(self.len, size=n)                              # Conv_model.py:147
        return idx, np.array(self.data_s)[idx], np.array(self.data_s_)[idx]


class PixelMemory(FrameMemory):
    """All nine fields of `Conv_model.Memory` (Conv_model.py:102-157); `sample` returns them in the reference's order,
    which names `b_lstm_s` twice (the stored `lstm_s_` never comes back, Conv_model.py:155-157).  Pinned by
    tests/golden/pixel_memory_full.npz (tests/golden/gen_pixel_memory_full_fixture.py)."""
    FIELDS = ("g", "lstm_s", "lstm_s_", "a", "dense_a", "r", "curious_r")

    def __init__(self, capacity: int):
        super().__init__(capacity)
        self.cols = {k: [] for k in self.FIELDS}

    def store_full(self, s_u8, g, lstm_s, a, r, curious_r, s_next_u8, lstm_s_, dense_a) -> None:
        vals = dict(g=g, lstm_s=lstm_s, lstm_s_=lstm_s_, a=a, dense_a=dense_a, r=r, curious_r=curious_r)
        if self.pointer > self.capacity:
            i = self.pointer % self.capacity
            for k, v in vals.items():
                self.cols[k][i] = v
        else:
            for k, v in vals.items():
                self.cols[k].append(v)
        self.store_transition(s_u8, s_next_u8)

    def sample_full(self, n: int, rs=np.random):
        idx = rs.choice(self.len, size=n)
        c = {k: np.array(v)[idx] for k, v in self.cols.items()}
        lstm = np.squeeze(c["lstm_s"], 2)
        return (np.array(self.data_s)[idx], c["g"], lstm, c["a"], c["r"], c["curious_r"], np.array(self.data_s_)[idx], lstm, c["dense_a"])
