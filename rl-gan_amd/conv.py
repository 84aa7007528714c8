"""Pixel feature extractors of the reference on the device: stacks of NHWC 'valid' convolutions with
ELU (`doom_feature_abstraction` Conv_model.py:15-24 / Conv_DDPG.py:83-86, `universe_feature_abstraction`
Conv_model.py:36-56 without batch norm).  Forward and weight gradients run as MFMA implicit GEMMs,
the input gradient as a deterministic gather (librlgan_amd.so, rlgan_conv_*)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .core import _stream, device

DOOM_HEAD = ((8, 5, 4), (16, 3, 2), (32, 3, 2), (64, 3, 2))     # (filters, kernel, stride), Conv_model.py:15-22


def universe_head(n_convs: int = 4):
    return tuple((32, 3, 2) for _ in range(n_convs))            # Conv_model.py:36-49


class FrameRef:
    """A minibatch of frames named by slot indices into a uint8 replay ring: `ring` uint8 [slots, h*w*c],
    `idx` int64 [n] on the device (None: slots 0..n-1), value = byte * scale (env_wrapper.py:44,58)."""

    def __init__(self, ring: torch.Tensor, idx: Optional[torch.Tensor], scale: float = 1.0 / 255, n: Optional[int] = None):
        self.ring, self.idx, self.scale = ring, idx, float(scale)
        self.n = int(idx.numel() if idx is not None else (ring.shape[0] if n is None else n))
        assert idx is None or (idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous())


class ConvLayer:
    """One conv2d + ELU layer: canonical parameters (HWIO kernel, then bias) and their fragment image."""

    def __init__(self, h: int, w: int, c_in: int, c_out: int, k: int, stride: int, rng: Optional[np.random.RandomState] = None,
                 bias_init: float = 0.0, kernel: Optional[np.ndarray] = None, bias: Optional[np.ndarray] = None):
        self.conv = _lib.Conv(h, w, c_in, c_out, k, stride)
        self.geom = _lib.ConvGeom()
        check(lib.rlgan_conv_geometry(self.conv, C.byref(self.geom)), "rlgan_conv_geometry")
        g = self.geom
        self.h_out, self.w_out, self.c_out, self.c_in, self.k, self.stride = g.h_out, g.w_out, c_out, c_in, k, stride
        self.n_params = g.n_params
        if kernel is None:
            rng = rng or np.random.RandomState()
            lim = np.sqrt(6.0 / (k * k * c_in + k * k * c_out))                  # tf.layers.conv2d default: glorot uniform
            kernel = rng.uniform(-lim, lim, (k, k, c_in, c_out))
            bias = np.full(c_out, bias_init)
        flat = np.concatenate([np.asarray(kernel, np.float32).ravel(), np.asarray(bias, np.float32).ravel()])
        assert flat.size == self.n_params
        dev = device()
        m = np.empty(self.n_params, np.int32)
        check(lib.rlgan_conv_build_map(self.conv, m.ctypes.data), "rlgan_conv_build_map")
        self.map_f = torch.from_numpy(m).to(dev)
        self.canon = torch.from_numpy(flat).to(dev)
        self.wfrag = torch.zeros(g.wfrag_floats, dtype=torch.float32, device=dev)
        # second image for the input gradient: the kernel regrouped by stride phase (absent when the
        # layer is outside what that kernel covers; the gather kernel then takes over)
        self.map_dx = self.wdx = None
        if g.wdx_floats > 0:
            md = np.empty(self.n_params, np.int32)
            check(lib.rlgan_conv_build_dx_map(self.conv, md.ctypes.data), "rlgan_conv_build_dx_map")
            self.map_dx = torch.from_numpy(md).to(dev)
            self.wdx = torch.zeros(g.wdx_floats, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.gather_dx = False   # True: input gradient by the vector-ALU gather even where the MFMA form exists
        # partial weight-gradient images: one per workgroup of the reduction over pixels; layers with few patch
        # elements (at most 128 incl. the bias row, and 32 filters) take one per wave when the batch is large
        self.n_slabs = 1024
        self._small = k * k * c_in + 1 <= 128 and c_out <= 32
        self._slabs = None
        self.pack()

    def pack(self) -> None:
        check(lib.rlgan_pack_weights(self.canon.data_ptr(), self.n_params, self.map_f.data_ptr(), self.wfrag.data_ptr(),
                                     None if self.wdx is None else self.map_dx.data_ptr(),
                                     None if self.wdx is None else self.wdx.data_ptr(), _stream()), "rlgan_pack_weights")

    def _slab_buffer(self, n_images: int):
        """(buffer, slab count) for the weight gradient over n_images."""
        ns = self.n_slabs
        if self._small and n_images * self.h_out * self.w_out >= 4096 * 256:
            ns = max(ns, 4096)
        need = ns * self.geom.slab_floats
        if self._slabs is None or self._slabs.numel() < need:
            self._slabs = torch.empty(need, dtype=torch.float32, device=self.canon.device)
        return self._slabs, ns

    def kernel_bias(self) -> Tuple[np.ndarray, np.ndarray]:
        f = self.canon.cpu().numpy()
        n = self.k * self.k * self.c_in * self.c_out
        return f[:n].reshape(self.k, self.k, self.c_in, self.c_out), f[n:]

    def forward(self, x: torch.Tensor, act: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out: where the result goes (float32 [n, h_out, w_out, c_out], contiguous), e.g. rows of a larger batch's buffer."""
        n = x.shape[0]
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape[1:]) == (self.conv.h, self.conv.w, self.c_in)
        y = torch.empty((n, self.h_out, self.w_out, self.c_out), dtype=torch.float32, device=x.device) if out is None else out
        assert y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and tuple(y.shape) == (n, self.h_out, self.w_out, self.c_out)
        check(lib.rlgan_conv_forward(self.conv, x.data_ptr(), self.wfrag.data_ptr(), y.data_ptr(), n, 1 if act else 0, _stream()),
              "rlgan_conv_forward")
        return y

    def forward_frames(self, frames: "FrameRef", act: bool = True) -> torch.Tensor:
        """This layer on uint8 frames read in place from a replay ring (no float32 copy of the frames)."""
        ring, idx, scale = frames.ring, frames.idx, frames.scale
        n = frames.n
        assert ring.is_cuda and ring.dtype == torch.uint8 and ring.is_contiguous() and ring.shape[1] == self.conv.h * self.conv.w * self.c_in
        y = torch.empty((n, self.h_out, self.w_out, self.c_out), dtype=torch.float32, device=ring.device)
        check(lib.rlgan_conv_forward_u8(self.conv, ring.data_ptr(), ring.shape[0], None if idx is None else idx.data_ptr(), scale,
                                        self.wfrag.data_ptr(), y.data_ptr(), n, 1 if act else 0, _stream()), "rlgan_conv_forward_u8")
        return y

    def backward_frames(self, frames: "FrameRef", y: torch.Tensor, dy: torch.Tensor, act: bool = True, dy_is_dz: bool = False) -> None:
        """Weight gradient of a first layer whose input is the uint8 ring (fills self.grad; frames get no gradient).
        dy_is_dz: dy is dZ already (the layer above multiplied its input gradient by ELU'(y))."""
        ring, idx = frames.ring, frames.idx
        slabs, ns = self._slab_buffer(frames.n)
        check(lib.rlgan_conv_backward_weights_u8(self.conv, ring.data_ptr(), ring.shape[0], None if idx is None else idx.data_ptr(),
                                                 frames.scale, dy.data_ptr(), y.data_ptr() if act and not dy_is_dz else None, slabs.data_ptr(),
                                                 ns, self.grad.data_ptr(), frames.n, _stream()),
              "rlgan_conv_backward_weights_u8")

    def backward(self, x: torch.Tensor, y: torch.Tensor, dy: torch.Tensor, need_dx: bool = True, act: bool = True,
                 need_dw: bool = True, dy_is_dz: bool = False, dx_elu: bool = False):
        """Gradients of this layer: fills self.grad (kernel then bias) unless need_dw is False, returns dx or None.
        dy_is_dz: dy is dZ = dY * ELU'(y) already.  dx_elu: x is the ELU output of the layer below and the returned gradient
        is multiplied by ELU'(x), i.e. it is that layer's dZ (the pair saves both of that layer's backward kernels the read
        of its output)."""
        n = x.shape[0]
        act = act and not dy_is_dz
        use_mfma = self.wdx is not None and not self.gather_dx
        # ELU' is applied by the consumers as they read dy (no pass of its own) unless the gather kernel runs
        fuse = act and (use_mfma or (self.c_in == 4 and self.conv.stride <= 4) or not need_dx)   # four input channels at stride <= 4: a kernel of its own, ELU' included
        dz, yel = dy, (y.data_ptr() if fuse else None)
        if act and not fuse:
            dz = torch.empty_like(dy)
            check(lib.rlgan_elu_backward(dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), _stream()), "rlgan_elu_backward")
        if need_dw:
            slabs, ns = self._slab_buffer(n)
            check(lib.rlgan_conv_backward_weights(self.conv, x.data_ptr(), dz.data_ptr(), yel, slabs.data_ptr(), ns,
                                                  self.grad.data_ptr(), n, _stream()), "rlgan_conv_backward_weights")
        if not need_dx:
            return None
        dx = torch.empty_like(x)
        check(lib.rlgan_conv_backward_input_dz(self.conv, dz.data_ptr(), yel, self.canon.data_ptr(),
                                               self.wdx.data_ptr() if use_mfma else None, dx.data_ptr(), x.data_ptr() if dx_elu else None,
                                               n, _stream()), "rlgan_conv_backward_input_dz")
        return dx


class SingleFrameConv(ConvLayer):
    """First layer of an encoder of ONE-channel frames, `tf.layers.conv2d(s, 32, 3, strides=2)` on [n, h, w, 1]
    (`universe_feature_abstraction(single_S_, time_step=1)`, Conv_model.py:36-49, Curious_GAN_mario.py:313-314), run as a
    2x2 stride-1 convolution of the frame's 2x2 blocks ([n, h/2, w/2, 4], channel 2*dy + dx = frame[2Y + dy][2X + dx]):
    output pixel (Y, X) reads frame rows 2Y..2Y+2 and columns 2X..2X+2, which lie in blocks (Y..Y+1, X..X+1).  The block
    kernel [2, 2, 4, f] holds the nine taps and seven zeros (frame row or column 2Y+3 / 2X+3); those seven stay zero:
    their gradient entries are cleared after every weight-gradient pass.  Against padding the frame to four channels
    this is a quarter of the input bytes and one group of 16 patch elements instead of three.

    kernel: the reference's [k, k, 1, f] (k = 3, stride 2, even h and w)."""

    def __init__(self, h: int, w: int, c_out: int, k: int = 3, stride: int = 2, rng: Optional[np.random.RandomState] = None,
                 bias_init: float = 0.0, kernel: Optional[np.ndarray] = None, bias: Optional[np.ndarray] = None):
        assert (k, stride) == (3, 2) and h % 2 == 0 and w % 2 == 0, "3x3 stride-2 on frames of even size"
        if kernel is None:
            rng = rng or np.random.RandomState()
            lim = np.sqrt(6.0 / (k * k + k * k * c_out))                         # glorot uniform on the [3, 3, 1, f] kernel
            kernel = rng.uniform(-lim, lim, (k, k, 1, c_out))
            bias = np.full(c_out, bias_init)
        kernel = np.asarray(kernel, np.float64).reshape(3, 3, c_out)
        blocks, real = np.zeros((2, 2, 4, c_out)), np.zeros((2, 2, 4, c_out), dtype=bool)
        for ky in range(3):
            for kx in range(3):
                blocks[ky // 2, kx // 2, 2 * (ky % 2) + kx % 2] = kernel[ky, kx]
                real[ky // 2, kx // 2, 2 * (ky % 2) + kx % 2] = True
        super().__init__(h // 2, w // 2, 4, c_out, 2, 1, kernel=blocks, bias=bias)
        self.frame_h, self.frame_w = h, w
        self._real = real
        self.mask = torch.from_numpy(np.concatenate([real.ravel(), np.ones(c_out, dtype=bool)]).astype(np.float32)).to(device())

    def _blocks(self, t: torch.Tensor, to_blocks: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = t.shape[0]
        shape = (n, self.frame_h // 2, self.frame_w // 2, 4) if to_blocks else (n, self.frame_h, self.frame_w, 1)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=t.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == shape
        check(lib.rlgan_blocks2x2(t.data_ptr(), out.data_ptr(), n, self.frame_h, self.frame_w, 1 if to_blocks else 0, _stream()),
              "rlgan_blocks2x2")
        return out

    def regroup(self, frames: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[n, h, w, 1] frames -> the [n, h/2, w/2, 4] blocks this layer's forward and backward take."""
        assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous()
        assert tuple(frames.shape[1:]) == (self.frame_h, self.frame_w, 1)
        return self._blocks(frames, True, out)

    def ungroup(self, dblocks: torch.Tensor) -> torch.Tensor:
        """The input gradient in block form -> d(frames) [n, h, w, 1]."""
        return self._blocks(dblocks.contiguous(), False)

    def kernel_bias(self) -> Tuple[np.ndarray, np.ndarray]:
        """(kernel [3, 3, 1, f], bias) as the reference holds them."""
        f = self.canon.cpu().numpy()
        nk = 16 * self.c_out
        blocks = f[:nk].reshape(2, 2, 4, self.c_out)
        kern = np.zeros((3, 3, 1, self.c_out), dtype=f.dtype)
        for ky in range(3):
            for kx in range(3):
                kern[ky, kx, 0] = blocks[ky // 2, kx // 2, 2 * (ky % 2) + kx % 2]
        return kern, f[nk:]

    def _tap_index(self) -> np.ndarray:
        """Position in the flat block layout [2, 2, 4, f] + bias of every entry of the reference's flat [3, 3, 1, f] + bias."""
        f = self.c_out
        pos = np.arange(16 * f).reshape(2, 2, 4, f)
        taps = [pos[ky // 2, kx // 2, 2 * (ky % 2) + kx % 2] for ky in range(3) for kx in range(3)]
        return np.concatenate([np.concatenate(taps), 16 * f + np.arange(f)])

    def to_frame_layout(self, t: torch.Tensor) -> np.ndarray:
        """A parameter-shaped vector of this layer (canon, gradient, optimizer slot) as the reference lays it out:
        kernel [3, 3, 1, f] flat, then the bias: 10 f numbers."""
        return t.detach().cpu().numpy()[self._tap_index()]

    def from_frame_layout(self, a: np.ndarray) -> torch.Tensor:
        """The inverse of `to_frame_layout`; the seven block entries outside the 3x3 window are zero."""
        a = np.asarray(a)
        assert a.shape == (10 * self.c_out,), a.shape
        out = np.zeros(17 * self.c_out, dtype=a.dtype)
        out[self._tap_index()] = a
        return torch.from_numpy(out)

    def frame_kernel_grad(self) -> Tuple[np.ndarray, np.ndarray]:
        """self.grad in the reference's shapes: (d kernel [3, 3, 1, f], d bias)."""
        g = self.grad.cpu().numpy()
        nk = 16 * self.c_out
        blocks = g[:nk].reshape(2, 2, 4, self.c_out)
        assert not blocks[~self._real].any()
        kern = np.zeros((3, 3, 1, self.c_out), dtype=g.dtype)
        for ky in range(3):
            for kx in range(3):
                kern[ky, kx, 0] = blocks[ky // 2, kx // 2, 2 * (ky % 2) + kx % 2]
        return kern, g[nk:]

    def backward(self, x, y, dy, need_dx: bool = True, act: bool = True, need_dw: bool = True, dy_is_dz: bool = False,
                 dx_elu: bool = False):
        dx = super().backward(x, y, dy, need_dx=need_dx, act=act, need_dw=need_dw, dy_is_dz=dy_is_dz, dx_elu=dx_elu)
        if need_dw:
            self.grad.mul_(self.mask)                          # the seven entries outside the 3x3 window take no gradient
        return dx


class DenseLayer:
    """The fully connected head of the encoders, `tf.layers.dense(features, 256, activation=tf.nn.elu)`
    (Conv_model.py:24 `fc`; Conv_model.py:52-56 `dense1` + elu).  Parameters in canonical order: the
    [K, N] kernel, then the bias row."""

    def __init__(self, k: int, n: int = 256, rng: Optional[np.random.RandomState] = None, bias_init: float = 0.0,
                 kernel: Optional[np.ndarray] = None, bias: Optional[np.ndarray] = None):
        assert k % 4 == 0 and n % 4 == 0, "16-byte rows"
        self.k, self.n, self.n_params = k, n, (k + 1) * n
        if kernel is None:
            rng = rng or np.random.RandomState()
            lim = np.sqrt(6.0 / (k + n))                                          # tf.layers.dense default: glorot uniform
            kernel = rng.uniform(-lim, lim, (k, n))
            bias = np.full(n, bias_init)
        flat = np.concatenate([np.asarray(kernel, np.float32).ravel(), np.asarray(bias, np.float32).ravel()])
        assert flat.size == self.n_params
        self.canon = torch.from_numpy(flat).to(device())
        self.grad = torch.zeros(self.n_params, dtype=torch.float32, device=device())
        self._scratch: Optional[torch.Tensor] = None

    def _slabs(self, m_out: int, k_red: int, n_out: Optional[int] = None):
        """(scratch, count) for a product with an m_out x n_out (default n) result summed over k_red terms."""
        n_out = self.n if n_out is None else n_out
        ns = lib.rlgan_dense_slabs(m_out, n_out, k_red)
        if ns <= 1:
            return None, 1
        need = ns * m_out * n_out
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float32, device=device())
        return self._scratch, ns

    def kernel_bias(self) -> Tuple[np.ndarray, np.ndarray]:
        f = self.canon.cpu().numpy()
        return f[:self.k * self.n].reshape(self.k, self.n), f[self.k * self.n:]

    @staticmethod
    def _act_code(act) -> int:
        """True / "elu": ELU (the encoders' head); "relu": ReLU (the pixel generator's first layer); "relu6" and "tanh":
        the pixel actor and critic; False / None: linear."""
        return {True: 1, "elu": 1, "relu": 2, "relu6": 3, "tanh": 4, False: 0, None: 0}[act]

    def forward(self, x: torch.Tensor, act=True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 2 and x.shape[1] == self.k
        y = torch.empty((x.shape[0], self.n), dtype=torch.float32, device=x.device) if out is None else out
        assert y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and tuple(y.shape) == (x.shape[0], self.n)
        sc, ns = self._slabs(x.shape[0], self.k + 1)
        check(lib.rlgan_dense_forward(x.data_ptr(), self.k, self.canon.data_ptr(), y.data_ptr(), self.n, x.shape[0], self.k, self.n,
                                      self._act_code(act), None if sc is None else sc.data_ptr(), ns, _stream()), "rlgan_dense_forward")
        return y

    def backward(self, x: torch.Tensor, y: torch.Tensor, dy: torch.Tensor, need_dx: bool = True, act=True, need_dw: bool = True):
        """Gradients of this layer: fills self.grad (kernel then bias) unless need_dw is False, returns dx or None."""
        m = x.shape[0]
        dz = dy
        code = self._act_code(act)
        if code:
            dz = torch.empty_like(dy)
            if code >= 3:
                check(lib.rlgan_act_backward(dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), code, _stream()), "rlgan_act_backward")
            else:
                fn, name = (lib.rlgan_elu_backward, "rlgan_elu_backward") if code == 1 else (lib.rlgan_relu_backward, "rlgan_relu_backward")
                check(fn(dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), _stream()), name)
        if need_dw:
            sc, ns = self._slabs(self.k + 1, m)
            check(lib.rlgan_dense_backward_weights(x.data_ptr(), self.k, dz.data_ptr(), self.n, self.grad.data_ptr(), m, self.k, self.n,
                                                   None if sc is None else sc.data_ptr(), ns, _stream()), "rlgan_dense_backward_weights")
        if not need_dx:
            return None
        dx = torch.empty_like(x)
        sc, ns = self._slabs(m, self.n, self.k)
        check(lib.rlgan_dense_backward_input_split(dz.data_ptr(), self.n, self.canon.data_ptr(), dx.data_ptr(), self.k, m, self.k, self.n,
                                                   None if sc is None else sc.data_ptr(), ns, _stream()), "rlgan_dense_backward_input_split")
        return dx


class ConvEncoder:
    """A stack of ConvLayers on [n, h, w, c] frames; output flattened to [n, features] like the
    reference's `tf.reshape(s, [-1, prod(shape[1:])])` (Conv_model.py:23,51).  c_in = 1 (the single-frame encoder,
    Curious_GAN_mario.py:313-314): frames [n, h, w, 1] in, d(frames) [n, h, w, 1] out, a SingleFrameConv in front."""

    def __init__(self, h: int, w: int, c_in: int, layers: Sequence[Tuple[int, int, int]] = DOOM_HEAD,
                 rng: Optional[np.random.RandomState] = None, bias_init: float = 0.0, params=None,
                 dense: int = 0, dense_params=None):
        """dense > 0 appends the `dense`-unit ELU head (256 in the reference); its parameters are drawn
        after the convolutions' from the same stream, or given as (kernel, bias)."""
        self.layers: List[ConvLayer] = []
        rng = rng or np.random.RandomState()
        for i, (f, k, s) in enumerate(layers):
            kw = {} if params is None else {"kernel": params[i][0], "bias": params[i][1]}
            if i == 0 and c_in == 1:                            # one-channel frames: the first layer works on 2x2 blocks
                L = SingleFrameConv(h, w, f, k, s, rng=rng, bias_init=bias_init, **kw)
            else:
                L = ConvLayer(h, w, c_in, f, k, s, rng=rng, bias_init=bias_init, **kw)
            self.layers.append(L)
            h, w, c_in = L.h_out, L.w_out, f
        self.conv_features = h * w * c_in
        self.head: Optional[DenseLayer] = None
        if dense:
            kw = {} if dense_params is None else {"kernel": dense_params[0], "bias": dense_params[1]}
            self.head = DenseLayer(self.conv_features, dense, rng=rng, bias_init=bias_init, **kw)
        self.features = dense or self.conv_features
        self._acts: List[torch.Tensor] = []
        self._flat = self._out = None

    def forward(self, x) -> torch.Tensor:
        """x: float32 frames [n, h, w, c], or a FrameRef (uint8 frames read in place by the first layer)."""
        if isinstance(self.layers[0], SingleFrameConv):
            x = self.layers[0].regroup(x)
        self._acts = [x]
        for i, L in enumerate(self.layers):
            x = L.forward_frames(x) if i == 0 and isinstance(x, FrameRef) else L.forward(x)
            self._acts.append(x)
        flat = x.reshape(x.shape[0], -1)
        if self.head is None:
            return flat
        self._flat, self._out = flat, self.head.forward(flat)
        return self._out

    # ---- one batch encoded in several calls ----------------------------------------------------------------
    def new_activation_set(self, total: int) -> List[torch.Tensor]:
        """Buffers for the activations of `total` frames: the first layer's input, every layer's output and the head's.
        `forward_rows` fills row ranges of them in separate calls; `use` then makes them the pass `backward` continues from.
        For a batch whose parts become available at different times while the encoder's variables stay the same (the
        discriminator's real and generated frames, Conv_model.py:183-199: the real ones are already encoded for the
        generator's loss), and it keeps the buffers of a recorded graph in one place."""
        dev = self.layers[0].canon.device
        first = self.layers[0]
        sets = [torch.empty((total, first.conv.h, first.conv.w, first.c_in), dtype=torch.float32, device=dev)]
        for L in self.layers:
            sets.append(torch.empty((total, L.h_out, L.w_out, L.c_out), dtype=torch.float32, device=dev))
        if self.head is not None:
            sets.append(torch.empty((total, self.head.n), dtype=torch.float32, device=dev))
        return sets

    def forward_rows(self, x: torch.Tensor, acts: List[torch.Tensor], row0: int) -> torch.Tensor:
        """Encodes float32 frames x [n, h, w, c] into rows row0 .. row0 + n of the activation set; returns their features
        (a view of the set's last buffer).  A row's results do not depend on what else is in the batch."""
        n = x.shape[0]
        rows = slice(row0, row0 + n)
        assert 0 <= row0 and row0 + n <= acts[0].shape[0]
        if isinstance(self.layers[0], SingleFrameConv):
            self.layers[0].regroup(x, out=acts[0][rows])
        else:
            acts[0][rows].copy_(x)
        for i, L in enumerate(self.layers):
            L.forward(acts[i][rows], out=acts[i + 1][rows])
        conv_out = acts[len(self.layers)][rows]
        if self.head is None:
            return conv_out.reshape(n, -1)
        return self.head.forward(conv_out.reshape(n, -1), out=acts[-1][rows])

    def use(self, acts: List[torch.Tensor]) -> torch.Tensor:
        """Makes a filled activation set the encoder's latest pass (what `backward` differentiates); returns its features."""
        nl = len(self.layers)
        self._acts = list(acts[:nl + 1])
        flat = acts[nl].reshape(acts[nl].shape[0], -1)
        if self.head is None:
            return flat
        self._flat, self._out = flat, acts[-1]
        return self._out

    def backward(self, dfeat: torch.Tensor, need_dx: bool = False, need_dw: bool = True) -> Optional[torch.Tensor]:
        """Backpropagates d(loss)/d(features) through the stack; every layer's .grad is filled, or with need_dw False
        left as it is (a loss that only passes through the encoder, like the generator's: Conv_model.py:529-531)."""
        assert need_dw or need_dx
        # (the dense head handing its dZ to the top conv layer the same way -- an ELU' factor in the epilogue of its input
        # product -- was measured: 6.52 -> 6.55 ms for the pixel iteration, the 42 MB it saves each conv kernel cost the product more)
        if self.head is not None:
            dfeat = self.head.backward(self._flat, self._out, dfeat.contiguous(), need_dw=need_dw)
        dy = dfeat.reshape(self._acts[-1].shape).contiguous()
        # every layer below the top gets its dZ from the layer above (that layer's input gradient times ELU' of its own input)
        # and reads its output no more
        top = len(self.layers) - 1
        for i in range(top, -1, -1):
            if i == 0 and isinstance(self._acts[0], FrameRef):
                if need_dw:
                    self.layers[0].backward_frames(self._acts[0], self._acts[1], dy, dy_is_dz=i < top)
                return None
            dy = self.layers[i].backward(self._acts[i], self._acts[i + 1], dy, need_dx=need_dx or i > 0, need_dw=need_dw,
                                         dy_is_dz=i < top, dx_elu=i > 0)
        if need_dx and isinstance(self.layers[0], SingleFrameConv):
            dy = self.layers[0].ungroup(dy)
        return dy
