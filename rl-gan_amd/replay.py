"""HBM-resident replay ring and the bit-exact index sampler.

The reference keeps a float64 NumPy ring on the host and draws minibatch indices with
`np.random.choice(capacity, size=n)` from the global legacy stream (Curious_GAN.py:155-169,
DDPG.py:181-196).  Here the ring is an fp32 device array with 128-byte rows (the networks consume
float32 either way: TF casts on feed) and the index draw runs on the device on a uploaded copy of
the MT19937 state, reproducing NumPy's stream bit for bit.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ._lib import check, lib
from .core import _stream, device


class MT19937Device:
    """Legacy NumPy MT19937 stream living on the device: uint32 key[624] + position."""

    def __init__(self, seed: Optional[int] = None):
        self.state = torch.zeros(625, dtype=torch.int32, device=device())
        if seed is not None:
            self.seed(seed)

    def seed(self, seed: int) -> None:
        check(lib.rlgan_mt19937_seed(self.state.data_ptr(), int(seed) & 0xFFFFFFFF, _stream()), "rlgan_mt19937_seed")

    def set_numpy_state(self, st) -> None:
        name, key, pos = st[0], st[1], st[2]
        assert name == "MT19937"
        host = np.empty(625, dtype=np.uint32)
        host[:624] = key
        host[624] = pos
        self.state.copy_(torch.from_numpy(host.view(np.int32)))

    def get_numpy_state(self, has_gauss: int = 0, cached: float = 0.0):
        host = self.state.cpu().numpy().view(np.uint32)
        return ("MT19937", host[:624].copy(), int(host[624]), has_gauss, cached)

    def choice(self, n_pop: int, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Indices of RandomState.choice(n_pop, size=n), int64 on the device."""
        if out is None:
            out = torch.empty(n, dtype=torch.int64, device=self.state.device)
        check(lib.rlgan_mt19937_choice(self.state.data_ptr(), int(n_pop), int(n), out.data_ptr(), _stream()),
              "rlgan_mt19937_choice")
        return out


class DeviceMemory:
    """Ring buffer `[capacity, dims]` with the reference's `pointer / capacity / data / sample` API.

    rng: None -> draw indices from NumPy's GLOBAL legacy stream like the reference does (the state is
    uploaded, advanced on the device and written back, so host code sharing that stream stays in
    step); or an `MT19937Device` that keeps its own stream on the device (no host traffic).
    """

    def __init__(self, capacity: int, dims: int, rng: Optional[MT19937Device] = None):
        self.capacity, self.dims = int(capacity), int(dims)
        self.ld = max(32, (dims + 31) // 32 * 32)
        dev = device()
        self._data = torch.zeros((self.capacity, self.ld), dtype=torch.float32, device=dev)
        self.pointer = 0
        self.rng = rng
        self._global_rng = MT19937Device() if rng is None else None
        self._stage = torch.zeros((1, self.ld), dtype=torch.float32, device=dev)

    # ---- reference API ------------------------------------------------------------------------
    @property
    def data(self) -> np.ndarray:
        return self._data[:, :self.dims].cpu().numpy().astype(np.float64)

    def store_row(self, row) -> None:
        row = np.asarray(row, dtype=np.float32).reshape(-1)
        assert row.size == self.dims
        self._stage[0, :self.dims].copy_(torch.from_numpy(row))
        self.store_rows(self._stage[:, :self.dims])

    def store_rows(self, rows: torch.Tensor) -> None:
        """rows: [n, dims] fp32 device tensor; row i goes to slot (pointer + i) % capacity."""
        n = rows.shape[0]
        check(lib.rlgan_replay_store(self._data.data_ptr(), self.capacity, self.dims, self.ld, self.pointer,
                                     rows.data_ptr(), rows.stride(0), n, _stream()), "rlgan_replay_store")
        self.pointer += n

    def sample_indices(self, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert self.pointer >= self.capacity, 'Memory has not been fulfilled'   # Curious_GAN.py:167
        if self.rng is not None:
            return self.rng.choice(self.capacity, n, out)
        st = np.random.get_state()
        self._global_rng.set_numpy_state(st)
        idx = self._global_rng.choice(self.capacity, n, out)
        np.random.set_state(self._global_rng.get_numpy_state(st[3], st[4]))
        return idx

    def gather(self, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = idx.numel()
        if out is None:
            out = torch.empty((n, self.dims), dtype=torch.float32, device=self._data.device)
        check(lib.rlgan_replay_gather(self._data.data_ptr(), self.capacity, self.dims, self.ld, idx.data_ptr(), n,
                                      out.data_ptr(), out.stride(0), _stream()), "rlgan_replay_gather")
        return out

    def sample_device(self, n: int) -> torch.Tensor:
        return self.gather(self.sample_indices(n))

    def sample(self, n: int) -> np.ndarray:
        """[n, dims] float64 host array, like the reference's `data[indices, :]`.  The rows stay on the device as
        well: column slices of the returned array (`b_M[:, :state_dim]`, Curious_GAN.py:201-205) handed to a `learn`
        or `eval` call are recognised by `device_view` and served from there instead of being uploaded again.  The
        array is read-only so that those device rows cannot go stale."""
        global _LAST_SAMPLE
        dev = self.sample_device(n)
        host = dev.cpu().numpy().astype(np.float64)
        host.flags.writeable = False
        _LAST_SAMPLE = (host, dev)
        return host


_LAST_SAMPLE = None                                             # (host array, device rows) of the newest `sample` call


def device_view(x) -> Optional[torch.Tensor]:
    """The device copy of `x` when x is a column slice (all rows) of the newest sampled minibatch, else None."""
    if _LAST_SAMPLE is None or not isinstance(x, np.ndarray):
        return None
    host, dev = _LAST_SAMPLE
    if x is host:
        return dev
    if x.base is not host or x.ndim != 2 or x.shape[0] != host.shape[0] or x.strides != host.strides:
        return None
    c0 = (x.__array_interface__["data"][0] - host.__array_interface__["data"][0]) // host.itemsize
    return dev[:, c0:c0 + x.shape[1]]


class FrameMemory:
    """Replay of frame stacks for the pixel variants: the reference's `Conv_model.Memory` (Conv_model.py:102-157)
    for its image fields, with the frames kept as the uint8 stacks the environment wrapper scales by 1/255
    (env_wrapper.py:44,58) instead of float32 lists -- a quarter of the memory, and the gather restores the
    same float32 values bit for bit.  Slot rule of the reference: transitions are appended while
    `pointer <= capacity` (capacity + 1 slots exist) and then overwrite slot `pointer % capacity`;
    `sample` draws from `len` = min(pointer, capacity + 1) slots (Conv_model.py:119-147).

    rng: as for DeviceMemory (None: NumPy's global legacy stream, like the reference)."""

    SCALE = 1.0 / 255

    def __init__(self, capacity: int, shape, rng: Optional[MT19937Device] = None):
        self.capacity, self.shape = int(capacity), tuple(int(v) for v in shape)
        self.frame_bytes = int(np.prod(self.shape))
        assert self.frame_bytes % 4 == 0, "frame stacks are moved four pixels at a time"
        dev = device()
        self.slots = self.capacity + 1
        # both fields in one allocation, s' after s: slot i of s' is row slots + i, so one index array can name frames of both
        self._both = torch.zeros((2 * self.slots, self.frame_bytes), dtype=torch.uint8, device=dev)
        self._s, self._s_next = self._both[:self.slots], self._both[self.slots:]
        self.pointer = 0
        self.len = 0
        self.rng = rng
        self._global_rng = MT19937Device() if rng is None else None

    def _slots_for(self, n: int) -> torch.Tensor:
        p = np.arange(self.pointer, self.pointer + n, dtype=np.int64)
        return torch.from_numpy(np.where(p > self.capacity, p % self.capacity, p)).to(self._s.device)

    def store_transitions(self, s_u8, s_next_u8) -> None:
        """s_u8, s_next_u8: uint8 [n, *shape] (torch on the device, or NumPy): n consecutive transitions."""
        fr = []
        for f in (s_u8, s_next_u8):
            t = torch.from_numpy(np.ascontiguousarray(f)) if isinstance(f, np.ndarray) else f
            assert t.dtype == torch.uint8
            fr.append(t.to(self._s.device).reshape(-1, self.frame_bytes).contiguous())
        n = fr[0].shape[0]
        assert fr[1].shape[0] == n and n <= self.capacity, "one call must not write a slot twice"
        slots = self._slots_for(n)
        for ring, t in ((self._s, fr[0]), (self._s_next, fr[1])):
            check(lib.rlgan_frames_store(ring.data_ptr(), self.slots, self.frame_bytes, t.data_ptr(), n, slots.data_ptr(), _stream()),
                  "rlgan_frames_store")
        self.pointer += n
        self.len = min(self.pointer, self.slots)

    def store_transition(self, s_u8, s_next_u8) -> None:
        self.store_transitions(np.asarray(s_u8)[None], np.asarray(s_next_u8)[None])

    def sample_indices(self, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert self.len > 0
        if self.rng is not None:
            return self.rng.choice(self.len, n, out)
        st = np.random.get_state()
        self._global_rng.set_numpy_state(st)
        idx = self._global_rng.choice(self.len, n, out)
        np.random.set_state(self._global_rng.get_numpy_state(st[3], st[4]))
        return idx

    def gather(self, idx: torch.Tensor, which: str = "s", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float32 [n, *shape] = uint8 frames * (1/255) of the slots idx (int64, device)."""
        ring = self._s if which == "s" else self._s_next
        n = idx.numel()
        if out is None:
            out = torch.empty((n,) + self.shape, dtype=torch.float32, device=ring.device)
        check(lib.rlgan_frames_gather(ring.data_ptr(), self.slots, self.frame_bytes, idx.data_ptr(), n, self.SCALE, out.data_ptr(),
                                      _stream()), "rlgan_frames_gather")
        return out

    def ref(self, idx: torch.Tensor, which: str = "s"):
        """The frames of slots idx as a FrameRef: a conv encoder reads them in place, without the float32 copy."""
        from .conv import FrameRef
        return FrameRef(self._s if which == "s" else self._s_next, idx, self.SCALE)

    def ref_pair(self, idx: torch.Tensor, out: Optional[torch.Tensor] = None):
        """s of slots idx followed by s' of the same slots as ONE FrameRef of 2n frames (the inverse model encodes both,
        Conv_model.py:864-867).  out: int64 [2n] on the device for the joined index array (rewritten by every call, so a
        recorded graph that reads it follows the current idx)."""
        from .conv import FrameRef
        n = idx.numel()
        if out is None:
            out = torch.empty(2 * n, dtype=torch.int64, device=idx.device)
        out[:n].copy_(idx)
        torch.add(idx, self.slots, out=out[n:])
        return FrameRef(self._both, out, self.SCALE)

    def gather_channel(self, idx: torch.Tensor, channel: int = -1, which: str = "s_", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float32 [n, h, w, 1]: one channel of the frames of slots idx, `b_s_[:, :, :, [-1]]` of Curious_GAN_mario.py:92
        by default -- the same values as that slice of `gather`, without the other channels' float32 copies."""
        ring = self._s if which == "s" else self._s_next
        c = self.shape[-1]
        channel = channel % c
        n, pixels = idx.numel(), self.frame_bytes // c
        if out is None:
            out = torch.empty((n,) + self.shape[:-1] + (1,), dtype=torch.float32, device=ring.device)
        check(lib.rlgan_frames_gather_channel(ring.data_ptr(), self.slots, pixels, c, channel, idx.data_ptr(), n, self.SCALE,
                                              out.data_ptr(), _stream()), "rlgan_frames_gather_channel")
        return out

    def sample(self, n: int):
        """(indices, b_s, b_s_) on the device: Conv_model.py:145-157 for the two image fields."""
        idx = self.sample_indices(n)
        return idx, self.gather(idx, "s"), self.gather(idx, "s_")


class PixelMemory:
    """`Conv_model.Memory` (Conv_model.py:102-157) with all nine fields of a transition, as the pixel training loop
    fills and reads it (Curious_GAN_mario.py:75,86): the frame stacks s and s' in a uint8 `FrameMemory`, everything
    else in one float32 ring with the same slots, row = [g | lstm_s (c, h) | lstm_s_ | a | dense_a | r | curious_r].

    Frames may be handed over as uint8 or as the wrapper's float32 `uint8 * 1/255` (env_wrapper.py:44,58), which is
    turned back into the bytes it came from; anything else is refused, because it could not be restored bit for bit.
    `sample` returns the reference's nine arrays in its order, `b_lstm_s` twice like the reference
    (Conv_model.py:155-157); the action comes back as float32 (the reference hands back the integer one-hot rows it
    was given, which every consumer feeds to a float32 placeholder)."""

    def __init__(self, capacity: int, shape=(42, 42, 4), a_dim: int = 6, lstm_units: int = 256, g_shape=None,
                 rng: Optional[MT19937Device] = None, dense_dim: Optional[int] = None):
        """dense_dim: width of the `dense_a` field; the training loop stores `transfer_sparse_action(a)`, a vector as
        long as the action (Curious_GAN_mario.py:64,267-270), which is the default; 1 for a scalar."""
        self.frames = FrameMemory(capacity, shape, rng=rng)
        self.capacity, self.shape, self.a_dim, self.units = int(capacity), tuple(shape), int(a_dim), int(lstm_units)
        self.g_shape = tuple(g_shape) if g_shape is not None else (self.shape[0], self.shape[1], 1)
        ng, nl = int(np.prod(self.g_shape)), 2 * self.units
        self.off = dict(g=0, lstm_s=ng, lstm_s_=ng + nl, a=ng + 2 * nl, dense_a=ng + 2 * nl + self.a_dim)
        self.dense_dim = self.a_dim if dense_dim is None else int(dense_dim)
        self.off["r"] = self.off["dense_a"] + self.dense_dim
        self.off["curious_r"] = self.off["r"] + 1
        self.dims = self.off["curious_r"] + 1
        self.ld = (self.dims + 31) // 32 * 32
        self.slots = self.capacity + 1
        self._rows = torch.zeros((self.slots, self.ld), dtype=torch.float32, device=device())
        self._stage = torch.zeros((1, self.ld), dtype=torch.float32, device=device())

    pointer = property(lambda self: self.frames.pointer)
    len = property(lambda self: self.frames.len)

    @staticmethod
    def _bytes(f) -> np.ndarray:
        f = np.asarray(f)
        if f.dtype == np.uint8:
            return f
        u8 = np.rint(f.astype(np.float64) * 255.0).astype(np.uint8)
        if not np.array_equal(u8.astype(np.float32) * FrameMemory.SCALE, f.astype(np.float32)):
            raise ValueError("frame stacks must be uint8 or uint8 * (1/255) as float32 (env_wrapper.py:44,58)")
        return u8

    def store_transition(self, s, g, lstm_s, a, r, curious_r, s_, lstm_s_, dense_a) -> None:
        """Conv_model.py:116-143."""
        p = self.frames.pointer
        slot = p % self.capacity if p > self.capacity else p
        row = np.zeros(self.ld, dtype=np.float32)
        o = self.off
        row[o["g"]:o["lstm_s"]] = np.asarray(g, dtype=np.float32).reshape(-1)
        row[o["lstm_s"]:o["lstm_s_"]] = np.asarray(lstm_s, dtype=np.float32).reshape(-1)
        row[o["lstm_s_"]:o["a"]] = np.asarray(lstm_s_, dtype=np.float32).reshape(-1)
        row[o["a"]:o["dense_a"]] = np.asarray(a, dtype=np.float32).reshape(-1)
        row[o["dense_a"]:o["r"]] = np.asarray(dense_a, dtype=np.float32).reshape(-1)
        row[o["r"]] = float(np.ravel(r)[0])
        row[o["curious_r"]] = float(np.ravel(curious_r)[0])
        self._stage.copy_(torch.from_numpy(row)[None])
        # the float ring has capacity + 1 slots; with that as the ring size the store's `% size` leaves the slot alone
        check(lib.rlgan_replay_store(self._rows.data_ptr(), self.slots, self.dims, self.ld, slot, self._stage.data_ptr(), self.ld, 1,
                                     _stream()), "rlgan_replay_store")
        self.frames.store_transition(self._bytes(s), self._bytes(s_))

    def sample_device(self, n: int) -> dict:
        """The minibatch on the device: idx, s, s_ (float32 frames) and the float fields as column views of one
        gathered [n, dims] tensor."""
        idx = self.frames.sample_indices(n)
        rows = torch.empty((n, self.dims), dtype=torch.float32, device=self._rows.device)
        check(lib.rlgan_replay_gather(self._rows.data_ptr(), self.slots, self.dims, self.ld, idx.data_ptr(), n, rows.data_ptr(),
                                      rows.stride(0), _stream()), "rlgan_replay_gather")
        o = self.off
        cut = lambda a, b: rows[:, o[a]:(o[b] if b else self.dims)]
        return dict(idx=idx, rows=rows, s=self.frames.gather(idx, "s"), s_=self.frames.gather(idx, "s_"), g=cut("g", "lstm_s"),
                    lstm_s=cut("lstm_s", "lstm_s_"), lstm_s_=cut("lstm_s_", "a"), a=cut("a", "dense_a"), dense_a=cut("dense_a", "r"),
                    r=cut("r", "curious_r"), curious_r=cut("curious_r", None))

    def sample(self, n: int):
        """Conv_model.py:145-157: (b_s, b_g, b_lstm_s, b_a, b_r, b_curious_r, b_s_, b_lstm_s, b_d_a) as NumPy arrays."""
        d = self.sample_device(n)
        rows = d["rows"].cpu().numpy()
        o = self.off
        b_g = rows[:, o["g"]:o["lstm_s"]].reshape((n,) + self.g_shape)
        b_lstm = rows[:, o["lstm_s"]:o["lstm_s_"]].reshape(n, 2, self.units)
        return (d["s"].cpu().numpy(), b_g, b_lstm, rows[:, o["a"]:o["dense_a"]], rows[:, o["r"]].astype(np.float64),
                rows[:, o["curious_r"]:o["curious_r"] + 1], d["s_"].cpu().numpy(), b_lstm,
                rows[:, o["dense_a"]] if self.dense_dim == 1 else rows[:, o["dense_a"]:o["r"]])
