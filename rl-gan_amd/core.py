"""Device-resident network state and the update-step engine.

PyTorch is used for device memory, streams and torch.distributed only; every computation on the
hot path is a call into librlgan_amd.so.

A `DeviceNet` owns the three device images of one network (canonical flat parameters in
TensorFlow variable order, forward fragments, backward fragments) plus the index maps between
them.  `StepEngine` strings the library's kernels into the reference's four update steps and the
two post-update loss evaluations (Curious_GAN.py:208-213).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import Acts, Env, Mat, Net, check, lib

HIDDEN = 200  # DDPG.py:82,159 ; Curious_GAN.py:59,121


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("rl-gan_amd needs an AMD GPU (MI355X, gfx950); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def mat(t: Optional[torch.Tensor]) -> Mat:
    """[rows, w] fp32 device tensor (unit stride in the last dim) -> (pointer, row stride)."""
    if t is None:
        return Mat(None, 0)
    assert t.dtype == torch.float32 and t.is_cuda
    if t.dim() == 1:
        return Mat(t.data_ptr(), t.stride(0))
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), "last dimension must be contiguous"
    return Mat(t.data_ptr(), t.stride(0))


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------------
# tensor tables: (TF variable name, shape) in creation order, bias constants
# --------------------------------------------------------------------------------------------
def tensor_table(kind: int, s_dim: int, a_dim: int) -> Tuple[List[Tuple[str, Tuple[int, ...]]], Tuple[float, ...]]:
    H = HIDDEN
    if kind == _lib.NET_ACTOR:        # DDPG.py:80-93
        return ([("l1/kernel", (s_dim, H)), ("l1/bias", (H,)), ("l2/kernel", (H, H)), ("l2/bias", (H,)),
                 ("l3/kernel", (H, 10)), ("l3/bias", (10,)), ("a/a/kernel", (10, a_dim)), ("a/a/bias", (a_dim,))],
                (0.001, 0.001, 0.001, 0.0))
    if kind == _lib.NET_CRITIC:       # DDPG.py:155-171
        return ([("l1/w1_s", (s_dim, H)), ("l1/w1_a", (a_dim, H)), ("l1/b1", (1, H)), ("l2/kernel", (H, H)),
                 ("l2/bias", (H,)), ("l3/kernel", (H, 10)), ("l3/bias", (10,)), ("q/dense/kernel", (10, 1)),
                 ("q/dense/bias", (1,))], (0.01, 0.01, 0.01, 0.01))
    if kind == _lib.NET_GENERATOR:    # Curious_GAN.py:114-134
        return ([("l1/w1_s", (s_dim, H)), ("l1/w1_a", (a_dim, H)), ("l1/b1", (1, H)), ("l2/kernel", (H, H)),
                 ("l2/bias", (H,)), ("l3/kernel", (H, 50)), ("l3/bias", (50,)), ("l4/kernel", (50, s_dim)),
                 ("l4/bias", (s_dim,))], (0.001,) * 4)
    if kind == _lib.NET_DISCRIMINATOR:  # Curious_GAN.py:52-81
        return ([("l1/w1_G", (s_dim, H)), ("l1/w1_s", (s_dim, H)), ("l1/w1_a", (a_dim, H)), ("l1/b1", (1, H)),
                 ("l2/kernel", (H, H)), ("l2/bias", (H,)), ("l3/kernel", (H, 10)), ("l3/bias", (10,)),
                 ("D/w3", (10, 1)), ("D/b3", (1, 1))], (0.001,) * 4)
    if kind == _lib.NET_PIXEL_ACTOR:    # Conv_model.py:713-739; a_dim here = the action columns padded to a multiple of 4
        return ([("l1/kernel", (s_dim, H)), ("l1/bias", (H,)), ("l2/kernel", (H, H)), ("l2/bias", (H,)),
                 ("l3/kernel", (H, 12)), ("l3/bias", (12,)), ("a/kernel", (12, a_dim)), ("a/bias", (a_dim,))],
                (0.001, 0.001, 0.001, 0.0))
    if kind == _lib.NET_PIXEL_CRITIC:   # Conv_model.py:623-650; w1_s and w1_a as one block, q in column 0 of four
        return ([("l1/w1", (s_dim + a_dim, H)), ("l1/b1", (H,)), ("l2/kernel", (H, H)), ("l2/bias", (H,)),
                 ("l3/kernel", (H, 12)), ("l3/bias", (12,)), ("q/kernel", (12, 4)), ("q/bias", (4,))],
                (0.01, 0.01, 0.01, 0.01))
    raise ValueError(kind)


def _is_bias(name: str) -> bool:
    return name.split("/")[-1] in ("bias", "b1", "b3")


def _layer_of(name: str) -> int:
    return {"l1": 0, "l2": 1, "l3": 2}.get(name.split("/")[0], 3)


def xavier_init(kind: int, s_dim: int, a_dim: int, rng: np.random.RandomState) -> np.ndarray:
    """tf.contrib.layers.xavier_initializer() (uniform, limit sqrt(6/(fan_in+fan_out))) for the
    weights, the reference's constants for the biases (DDPG.py:80-81,92-93,155-156)."""
    table, bias = tensor_table(kind, s_dim, a_dim)
    parts = []
    for name, shape in table:
        if _is_bias(name):
            parts.append(np.full(int(np.prod(shape)), bias[_layer_of(name)], dtype=np.float32))
        else:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            parts.append(rng.uniform(-lim, lim, size=shape).astype(np.float32).reshape(-1))
    return np.concatenate(parts)


class DeviceNet:
    """One network on the device: canon / forward fragments / backward fragments + index maps."""

    def __init__(self, kind: int, s_dim: int, a_dim: int, trainable: bool = True,
                 init: Optional[np.ndarray] = None, rng: Optional[np.random.RandomState] = None):
        self.kind, self.s_dim, self.a_dim, self.trainable = kind, s_dim, a_dim, trainable
        self.geom = _lib.geometry(kind, s_dim, a_dim)
        n = self.geom.n_params
        self.table, _ = tensor_table(kind, s_dim, a_dim)
        assert sum(int(np.prod(s)) for _, s in self.table) == n
        mf = np.empty(n, np.int32)
        mb = np.empty(n, np.int32)
        mg = np.empty(n, np.int32)
        check(lib.rlgan_build_maps(kind, s_dim, a_dim, mf.ctypes.data, mb.ctypes.data, mg.ctypes.data), "rlgan_build_maps")
        dev = device()
        self.map_f = torch.from_numpy(mf).to(dev)
        self.map_b = torch.from_numpy(mb).to(dev) if trainable else None
        self.map_g = torch.from_numpy(mg).to(dev)
        self.canon = torch.zeros(n, dtype=torch.float32, device=dev)
        self.wf = torch.zeros(self.geom.wf_floats, dtype=torch.float32, device=dev)
        self.wb = torch.zeros(self.geom.wb_floats, dtype=torch.float32, device=dev) if trainable else None
        if init is None:
            init = xavier_init(kind, s_dim, a_dim, rng or np.random.RandomState())
        self.set_flat(init)

    # ---- views handed to the C ABI
    def net(self, backward: bool = False) -> Net:
        return Net(self.wf.data_ptr(), self.wb.data_ptr() if (backward and self.wb is not None) else None)

    @property
    def n_params(self) -> int:
        return self.geom.n_params

    # ---- parameter I/O (canonical layout)
    def set_flat(self, flat) -> None:
        flat = np.ascontiguousarray(np.asarray(flat, dtype=np.float32).reshape(-1))
        assert flat.size == self.n_params
        self.canon.copy_(torch.from_numpy(flat))
        self.repack()

    def repack(self) -> None:
        check(lib.rlgan_pack_weights(self.canon.data_ptr(), self.n_params, self.map_f.data_ptr(), self.wf.data_ptr(),
                                     ptr(self.map_b), ptr(self.wb), _stream()), "rlgan_pack_weights")

    def get_flat(self) -> np.ndarray:
        return self.canon.detach().cpu().numpy().copy()

    def get_tensors(self) -> "OrderedDict[str, np.ndarray]":
        flat, out, o = self.get_flat(), OrderedDict(), 0
        for name, shape in self.table:
            k = int(np.prod(shape))
            out[name] = flat[o:o + k].reshape(shape)
            o += k
        return out

    def set_tensors(self, tensors: Dict[str, np.ndarray]) -> None:
        self.set_flat(np.concatenate([np.asarray(tensors[n], dtype=np.float32).reshape(-1) for n, _ in self.table]))


class Workspace:
    """Work arrays one network needs for a training step over `rows` minibatch rows."""

    def __init__(self, net: DeviceNet, rows: int):
        dev = device()
        self.rows = rows
        g = net.geom
        f = lambda w: torch.zeros(rows * w, dtype=torch.float32, device=dev)
        x0_cols = 272 if net.kind >= _lib.NET_PIXEL_ACTOR else 32
        self.x0, self.h1, self.h2, self.h3 = f(x0_cols), f(208), f(208), f(g.ld3)
        self.dz1, self.dz2, self.dz3, self.dz4 = f(208), f(208), f(g.ld3), f(16)
        self.chunk_rows, self.n_chunks = _lib.wgrad_chunks(rows)
        self.gslabs = torch.zeros(self.n_chunks * g.gpad_floats, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(g.n_params, dtype=torch.float32, device=dev)

    def acts(self) -> Acts:
        return Acts(*[t.data_ptr() for t in (self.x0, self.h1, self.h2, self.h3, self.dz1, self.dz2, self.dz3, self.dz4)])


NULL_ACTS = Acts(*([None] * 8))
NULL_NET = Net(None, None)


def rows16(n: int) -> int:
    return (n + 15) // 16 * 16


def mlp_forward(net: DeviceNet, s: torch.Tensor, a: Optional[torch.Tensor] = None, x: Optional[torch.Tensor] = None,
                out_mode: int = 0) -> torch.Tensor:
    n = s.shape[0]
    out = torch.empty((n, net.geom.n_out), dtype=torch.float32, device=s.device)
    check(lib.rlgan_mlp_forward(net.kind, net.s_dim, net.a_dim, net.net(), mat(s), mat(x), mat(a), out.data_ptr(),
                                out.stride(0), out_mode, n, _stream()), "rlgan_mlp_forward")
    return out


def wgrad_job(net: DeviceNet, ws: Workspace) -> _lib.WgradJob:
    return _lib.WgradJob(net.kind, ws.rows, ws.acts(), ws.gslabs.data_ptr())


def reduce_job(net: DeviceNet, ws: Workspace, loss_parts=None, n_parts=0, loss_scale=0.0, loss_out=None,
               counter=None) -> _lib.ReduceJob:
    return _lib.ReduceJob(ws.gslabs.data_ptr(), ws.n_chunks, net.geom.gpad_floats, net.map_g.data_ptr(), net.n_params,
                          ws.grad.data_ptr(), ptr(loss_parts), n_parts, loss_scale, ptr(loss_out), ptr(counter))


def rmsprop_job(net: DeviceNet, grad: torch.Tensor, ms: torch.Tensor, lr: float, decay=0.9, eps=1e-10,
                target: Optional[DeviceNet] = None, tau: float = 1.0, counter: Optional[torch.Tensor] = None,
                period: int = 0) -> _lib.ApplyJob:
    """tf.train.RMSPropOptimizer step (DDPG.py:112,148), optionally followed by the target-net copy."""
    return _lib.ApplyJob(0, net.n_params, net.canon.data_ptr(), grad.data_ptr(), ms.data_ptr(), None,
                         lr, decay, 0.0, eps, None, net.map_f.data_ptr(), net.wf.data_ptr(), ptr(net.map_b), ptr(net.wb),
                         None if target is None else target.canon.data_ptr(),
                         None if target is None else target.wf.data_ptr(), tau, ptr(counter), period, None)


def adam_job(net: DeviceNet, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: torch.Tensor, lr: float,
             beta1=0.9, beta2=0.999, eps=1e-8) -> _lib.ApplyJob:
    """tf.train.AdamOptimizer step (Curious_GAN.py:45,142)."""
    return _lib.ApplyJob(1, net.n_params, net.canon.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(),
                         lr, beta1, beta2, eps, step.data_ptr(), net.map_f.data_ptr(), net.wf.data_ptr(),
                         ptr(net.map_b), ptr(net.wb), None, None, 1.0, None, 0, None)


def _arr(cls, jobs):
    return (cls * len(jobs))(*jobs)


def launch_wgrads(jobs, s_dim: int, a_dim: int) -> None:
    check(lib.rlgan_weight_grads(_arr(_lib.WgradJob, jobs), len(jobs), s_dim, a_dim, _stream()), "rlgan_weight_grads")


def launch_reduce(jobs) -> None:
    check(lib.rlgan_grad_reduce(_arr(_lib.ReduceJob, jobs), len(jobs), _stream()), "rlgan_grad_reduce")


def row_ld(t: torch.Tensor) -> int:
    """Leading dimension of a 2-D row tensor with unit column stride; a one-row tensor may carry any row stride."""
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def launch_apply(jobs, reduces=None) -> None:
    """Optimizer steps; with `reduces` (one ReduceJob per apply job) the chunk reduction of the weight
    gradients runs inside the same launch."""
    if reduces is not None:
        assert len(reduces) == len(jobs)
        for j, r in zip(jobs, reduces):
            j.reduce = C.pointer(r)
    check(lib.rlgan_apply(_arr(_lib.ApplyJob, jobs), len(jobs), _stream()), "rlgan_apply")


def weight_grads(net: DeviceNet, ws: Workspace, loss_parts=None, n_parts=0, loss_scale=0.0, loss_out=None,
                 counter=None) -> None:
    """dW for all layers of `net` from the work arrays, reduced over batch chunks into ws.grad."""
    launch_wgrads([wgrad_job(net, ws)], net.s_dim, net.a_dim)
    launch_reduce([reduce_job(net, ws, loss_parts, n_parts, loss_scale, loss_out, counter)])


def sum_parts(parts: torch.Tensor, n_parts: int, n_sums: int, scale: float, out: torch.Tensor) -> None:
    check(lib.rlgan_sum_parts(parts.data_ptr(), n_parts, n_sums, scale, out.data_ptr(), _stream()), "rlgan_sum_parts")


def target_update(target: DeviceNet, evalnet: DeviceNet, tau: float, counter: Optional[torch.Tensor], period: int) -> None:
    check(lib.rlgan_target_update(target.canon.data_ptr(), evalnet.canon.data_ptr(), target.n_params, tau,
                                  ptr(counter), period, target.map_f.data_ptr(), target.wf.data_ptr(), _stream()),
          "rlgan_target_update")
