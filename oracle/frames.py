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
