"""Decoder of the reference's full-prediction pixel generator on the device: a stack of NHWC 'valid'
transposed convolutions, 1x1x200 -> 3x3x128 -> 8x8x64 -> 19x19x32 -> 42x42x1, ReLU between the layers
(`Generator._build_full_generator`, Conv_model.py:457-473).  A layer is one MFMA matrix product with every
input pixel as a row, then a placement kernel that gathers the taps landing on each output pixel
(librlgan_amd.so, rlgan_deconv_* and rlgan_dense_*)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .core import _stream, device

DECODER = ((128, 3, 2), (64, 4, 2), (32, 5, 2), (1, 6, 2))      # (filters, kernel, stride), Conv_model.py:458-473


class DeconvLayer:
    """One conv2d_transpose (+ ReLU) layer.  Parameters in canonical order: TensorFlow's kernel
    [k, k, c_out, c_in], then the bias."""

    cols_view = True      # backward products read dz through a view of the cols matrix where the extents allow it (tests turn it off to compare)

    def __init__(self, h: int, w: int, c_in: int, c_out: int, k: int, stride: int, rng: Optional[np.random.RandomState] = None,
                 bias_init: float = 0.0, kernel: Optional[np.ndarray] = None, bias: Optional[np.ndarray] = None, fused: bool = True):
        """fused=False keeps the cols form for a shape that has a fused kernel (the two forms are compared in the tests)."""
        self.spec = _lib.Deconv(h, w, c_in, c_out, k, stride)
        self.geom = _lib.DeconvGeom()
        check(lib.rlgan_deconv_geometry(self.spec, C.byref(self.geom)), "rlgan_deconv_geometry")
        g = self.geom
        self.h, self.w, self.c_in, self.c_out, self.k, self.stride = h, w, c_in, c_out, k, stride
        self.h_out, self.w_out, self.kk, self.n_params = g.h_out, g.w_out, g.kk, g.n_params
        if kernel is None:
            rng = rng or np.random.RandomState()
            lim = np.sqrt(6.0 / (k * k * c_in + k * k * c_out))               # tf.layers default: glorot uniform
            kernel = rng.uniform(-lim, lim, (k, k, c_out, c_in))
            bias = np.full(c_out, bias_init)
        flat = np.concatenate([np.asarray(kernel, np.float32).ravel(), np.asarray(bias, np.float32).ravel()])
        assert flat.size == self.n_params
        dev = device()
        m = np.empty(self.n_params, np.int32)
        check(lib.rlgan_deconv_build_map(self.spec, m.ctypes.data), "rlgan_deconv_build_map")
        self.map_w = torch.from_numpy(m).to(dev)
        self.canon = torch.from_numpy(flat).to(dev)
        self.wmat = torch.zeros(g.wmat_floats, dtype=torch.float32, device=dev)     # [c_in + 1][kk], last row stays zero
        # the two decoder shapes with fused kernels read the kernel from two fragment images instead of wmat
        fg = _lib.DeconvFusedGeom()
        check(lib.rlgan_deconv_fused_geometry(self.spec, C.byref(fg)), "rlgan_deconv_fused_geometry")
        self.fused = fg.kind != 0 and fused                  # forward pass by a fused kernel
        self.fused_bwd = self.fused and fg.kind != 3          # kind 3: forward only, the backward pass keeps the cols form
        self.hands_dz = self.fused_bwd                        # its backward kernels can return dx * ReLU'(x): the dZ of the layer below
        if self.fused:
            mf, mb = np.empty(self.n_params, np.int32), np.empty(self.n_params, np.int32)
            check(lib.rlgan_deconv_fused_build_maps(self.spec, mf.ctypes.data, mb.ctypes.data if self.fused_bwd else None),
                  "rlgan_deconv_fused_build_maps")
            self.map_f = torch.from_numpy(mf).to(dev)
            self.wfrag = torch.zeros(fg.wff_floats, dtype=torch.float32, device=dev)
            if self.fused_bwd:
                self.map_dx = torch.from_numpy(mb).to(dev)
                self.wdx = torch.zeros(fg.wfb_floats, dtype=torch.float32, device=dev)
            else:                                             # the optimizer's second image is Wmat, which the backward pass reads
                self.map_dx, self.wdx = self.map_w, self.wmat
            self._ws: Optional[torch.Tensor] = None
        self.grad = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self._gmat = torch.empty(g.wmat_floats, dtype=torch.float32, device=dev)
        self._colsum = torch.empty(max(1, g.colsum_floats), dtype=torch.float32, device=dev)
        self._scratch: Optional[torch.Tensor] = None
        self._cols: Optional[torch.Tensor] = None
        self.pack()

    def pack(self) -> None:
        if self.fused:
            check(lib.rlgan_pack_weights(self.canon.data_ptr(), self.kk * self.c_in, self.map_f.data_ptr(), self.wfrag.data_ptr(),
                                         self.map_dx.data_ptr(), self.wdx.data_ptr(), _stream()), "rlgan_pack_weights")
            return
        check(lib.rlgan_pack_weights(self.canon.data_ptr(), self.kk * self.c_in, self.map_w.data_ptr(), self.wmat.data_ptr(), None, None,
                                     _stream()), "rlgan_pack_weights")

    def kernel_bias(self) -> Tuple[np.ndarray, np.ndarray]:
        f = self.canon.cpu().numpy()
        n = self.kk * self.c_in
        return f[:n].reshape(self.k, self.k, self.c_out, self.c_in), f[n:]

    @property
    def bias(self) -> torch.Tensor:
        return self.canon[self.kk * self.c_in:]

    def _slabs(self, m_out: int, n_out: int, k_red: int):
        ns = lib.rlgan_dense_slabs(m_out, n_out, k_red)
        if ns <= 1:
            return None, 1
        need = ns * m_out * n_out
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float32, device=device())
        return self._scratch, ns

    def _cols_buffer(self, rows: int) -> torch.Tensor:
        if self._cols is None or self._cols.numel() < rows * self.kk:
            self._cols = torch.empty(rows * self.kk, dtype=torch.float32, device=device())
        return self._cols

    def forward(self, x: torch.Tensor, act: bool = True) -> torch.Tensor:
        n = x.shape[0]
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape[1:]) == (self.h, self.w, self.c_in)
        if self.fused:
            y = torch.empty((n, self.h_out, self.w_out, self.c_out), dtype=torch.float32, device=x.device)
            check(lib.rlgan_deconv_fused_forward(self.spec, x.data_ptr(), self.wfrag.data_ptr(), self.bias.data_ptr(), y.data_ptr(), n,
                                                 1 if act else 0, _stream()), "rlgan_deconv_fused_forward")
            return y
        rows = n * self.h * self.w
        cols = self._cols_buffer(rows)
        sc, ns = self._slabs(rows, self.kk, self.c_in + 1)
        check(lib.rlgan_dense_forward(x.data_ptr(), self.c_in, self.wmat.data_ptr(), cols.data_ptr(), self.kk, rows, self.c_in, self.kk,
                                      0, None if sc is None else sc.data_ptr(), ns, _stream()), "rlgan_dense_forward")
        y = torch.empty((n, self.h_out, self.w_out, self.c_out), dtype=torch.float32, device=x.device)
        check(lib.rlgan_deconv_place(self.spec, cols.data_ptr(), self.bias.data_ptr(), y.data_ptr(), n, 1 if act else 0, _stream()),
              "rlgan_deconv_place")
        return y

    def backward(self, x: torch.Tensor, y: torch.Tensor, dy: torch.Tensor, need_dx: bool = True, act: bool = True,
                 dx_relu: bool = False):
        """Gradients of this layer: fills self.grad (kernel then bias), returns dx or None.  dx_relu (see `hands_dz`): the
        returned gradient is multiplied by ReLU'(x), i.e. it is the dZ of the layer below, which is then called with
        act=False."""
        n = x.shape[0]
        rows = n * self.h * self.w
        yr = y.data_ptr() if act else None
        if self.fused_bwd:
            need = lib.rlgan_deconv_fused_workspace(self.spec, n)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.float32, device=x.device)
            dx = torch.empty_like(x) if need_dx else None
            assert not dx_relu or (self.hands_dz and dx is not None)
            check(lib.rlgan_deconv_fused_backward_dz(self.spec, x.data_ptr(), yr, dy.data_ptr(), self.wdx.data_ptr(),
                                                     None if dx is None else dx.data_ptr(), 1 if dx_relu else 0, self._ws.data_ptr(),
                                                     self.grad.data_ptr(), n, _stream()), "rlgan_deconv_fused_backward_dz")
            return dx
        assert not dx_relu
        # dcols[(n, iy, ix)][(ky, kx, co)] is an element of dz = dy * ReLU'(y): where the layer's extents are whole tiles the
        # two products below read dz in place and the matrix is never written
        view = self.cols_view and lib.rlgan_deconv_view_ok(self.spec, n) == 1
        if view:
            if act:
                dz = torch.empty_like(dy)
                check(lib.rlgan_relu_backward(dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), _stream()), "rlgan_relu_backward")
            else:
                dz = dy
        else:
            dcols = self._cols_buffer(rows)
            check(lib.rlgan_deconv_gather(self.spec, dy.data_ptr(), yr, dcols.data_ptr(), n, _stream()), "rlgan_deconv_gather")
        sc, ns = self._slabs(self.c_in, self.kk, rows)            # the kernel gradient alone: the bias gradient is a column sum of dy (below)
        if view:
            check(lib.rlgan_deconv_backward_kernel_view(self.spec, x.data_ptr(), dz.data_ptr(), self._gmat.data_ptr(), n,
                                                        None if sc is None else sc.data_ptr(), ns, _stream()),
                  "rlgan_deconv_backward_kernel_view")
        else:
            check(lib.rlgan_dense_backward_kernel(x.data_ptr(), self.c_in, dcols.data_ptr(), self.kk, self._gmat.data_ptr(), rows, self.c_in,
                                                  self.kk, None if sc is None else sc.data_ptr(), ns, _stream()),
                  "rlgan_dense_backward_kernel")
        nk = self.kk * self.c_in
        check(lib.rlgan_gather_map(self._gmat.data_ptr(), self.map_w.data_ptr(), nk, self.grad.data_ptr(), _stream()), "rlgan_gather_map")
        # (through the view dz holds dy * ReLU'(y) already: the same values, half the reads)
        check(lib.rlgan_column_sums(dz.data_ptr() if view else dy.data_ptr(), None if view else yr, n * self.h_out * self.w_out, self.c_out,
                                    self._colsum.data_ptr(), self.grad[nk:].data_ptr(), _stream()), "rlgan_column_sums")
        if not need_dx:
            return None
        dx = torch.empty_like(x)
        sc, ns = self._slabs(rows, self.c_in, self.kk)
        if view:
            check(lib.rlgan_deconv_backward_input_view(self.spec, dz.data_ptr(), self.wmat.data_ptr(), dx.data_ptr(), n,
                                                       None if sc is None else sc.data_ptr(), ns, _stream()),
                  "rlgan_deconv_backward_input_view")
            return dx
        check(lib.rlgan_dense_backward_input_split(dcols.data_ptr(), self.kk, self.wmat.data_ptr(), dx.data_ptr(), self.c_in, rows, self.c_in,
                                                   self.kk, None if sc is None else sc.data_ptr(), ns, _stream()),
              "rlgan_dense_backward_input_split")
        return dx


class Decoder:
    """The decoder stack on [n, c] codes (the reference reshapes them to [n, 1, 1, c], Conv_model.py:457);
    ReLU after every layer but the last (Conv_model.py:458-473, batch norm off as in Curious_GAN_mario.py:384)."""

    def __init__(self, c_in: int = 200, layers: Sequence[Tuple[int, int, int]] = DECODER, rng: Optional[np.random.RandomState] = None,
                 bias_init: float = 0.0, params=None):
        self.layers: List[DeconvLayer] = []
        rng = rng or np.random.RandomState()
        h = w = 1
        self.c_in = c_in
        for i, (f, k, s) in enumerate(layers):
            kw = {} if params is None else {"kernel": params[i][0], "bias": params[i][1]}
            L = DeconvLayer(h, w, c_in, f, k, s, rng=rng, bias_init=bias_init, **kw)
            self.layers.append(L)
            h, w, c_in = L.h_out, L.w_out, f
        self.out_shape = (h, w, c_in)
        self._acts: List[torch.Tensor] = []

    def forward(self, code: torch.Tensor) -> torch.Tensor:
        x = code.reshape(code.shape[0], 1, 1, self.c_in)
        self._acts = [x]
        for i, L in enumerate(self.layers):
            x = L.forward(x, act=i + 1 < len(self.layers))
            self._acts.append(x)
        return x

    def backward(self, dimage: torch.Tensor, need_dx: bool = True) -> Optional[torch.Tensor]:
        """Backpropagates d(loss)/d(image); every layer's .grad is filled; returns d(loss)/d(code) [n, c]."""
        dy = dimage.contiguous()
        last = len(self.layers) - 1
        is_dz = False                                             # dy is dZ already: the layer above applied this layer's ReLU'
        for i in range(last, -1, -1):
            L = self.layers[i]
            hand = i > 0 and L.hands_dz                           # layer i - 1 has a ReLU (every layer but the last does)
            dy = L.backward(self._acts[i], self._acts[i + 1], dy, need_dx=need_dx or i > 0, act=i < last and not is_dz, dx_relu=hand)
            is_dz = hand
        return None if dy is None else dy.reshape(dy.shape[0], -1)
