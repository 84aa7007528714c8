"""The 2-link arm environment on the device (reference: GAN-RL/arm_env.py:17-115).

`ArmEnv` keeps the reference's scalar API (one environment, NumPy in / NumPy out, random draws taken
from NumPy's global stream exactly where the reference takes them); `VecArmEnv` is the batched form
the rollout uses.  Rendering (arm_env.py:118-215) is not part of this package.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ._lib import Env, check, lib
from .core import _stream, device


class VecArmEnv:
    """n environments, struct of arrays in HBM, float64 dynamics."""
    state_dim, action_dim, action_bound = 7, 2, [-1, 1]

    def __init__(self, n: int, mode: str = 'hard'):
        assert mode in ('easy', 'hard')
        dev = device()
        self.n, self.mode = n, mode
        self.theta = torch.zeros(n * 2, dtype=torch.float64, device=dev)     # arm_info[:, 1]
        self.ends = torch.zeros(n * 4, dtype=torch.float64, device=dev)      # arm_info[:, 2:4]  (0,0) until first step
        self.goal = torch.tensor([250.0, 303.0], dtype=torch.float64, device=dev).repeat(n)   # point_info (int-valued)
        self.grab = torch.zeros(n, dtype=torch.int32, device=dev)
        self.got = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.state64 = torch.zeros((n, 7), dtype=torch.float64, device=dev)
        self.state32 = torch.zeros((n, 7), dtype=torch.float32, device=dev)
        self.reward = torch.zeros(n, dtype=torch.float64, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)

    def handle(self) -> Env:
        return Env(self.theta.data_ptr(), self.ends.data_ptr(), self.goal.data_ptr(), self.grab.data_ptr(), self.got.data_ptr())

    def reset(self, u: torch.Tensor, mask: Optional[torch.Tensor] = None):
        """u: [n,2] float64 uniforms in [0,1) (np.random.rand(2) per environment, arm_env.py:66,69)."""
        assert u.dtype == torch.float64 and u.is_cuda and u.numel() == 2 * self.n
        check(lib.rlgan_env_reset(self.handle(), self.n, 1 if self.mode == 'hard' else 0,
                                  None if mask is None else mask.data_ptr(), u.data_ptr(),
                                  self.state64.data_ptr(), self.state32.data_ptr(), _stream()), "rlgan_env_reset")
        return self.state64

    def step(self, action: torch.Tensor):
        """action: [n,2] float32 or float64 device tensor.  Returns (state64, reward, done) device tensors."""
        assert action.is_cuda and action.shape == (self.n, 2) and action.stride(1) == 1
        f64 = 1 if action.dtype == torch.float64 else 0
        assert f64 or action.dtype == torch.float32
        check(lib.rlgan_env_step(self.handle(), self.n, action.data_ptr(), f64, action.stride(0),
                                 self.state64.data_ptr(), self.state32.data_ptr(), self.reward.data_ptr(),
                                 self.done.data_ptr(), _stream()), "rlgan_env_step")
        return self.state64, self.reward, self.done


class ArmEnv(object):
    """Drop-in for the reference class (arm_env.py:17-115)."""
    action_bound = [-1, 1]
    action_dim = 2
    state_dim = 7
    dt = .1
    viewer = None

    def __init__(self, mode='easy'):
        self.mode = mode
        self._v = VecArmEnv(1, mode)
        # state, reward and done of the one environment side by side, so that a step reads them back in one transfer
        self._out = torch.zeros(80, dtype=torch.uint8, device=device())
        self._v.state64 = self._out[:56].view(torch.float64).view(1, 7)
        self._v.reward = self._out[56:64].view(torch.float64)
        self._v.done = self._out[64:65]

    # state mirrors of the reference attributes
    @property
    def point_info(self):
        return self._v.goal.cpu().numpy().astype(np.int64)

    @property
    def arm_info(self):
        out = np.zeros((2, 4))
        out[:, 0] = 100
        out[:, 1] = self._v.theta.cpu().numpy()
        out[:, 2:4] = self._v.ends.cpu().numpy().reshape(2, 2)
        return out

    @property
    def grab_counter(self):
        return int(self._v.grab.item())

    @property
    def get_point(self):
        return bool(self._v.got.item())

    def step(self, action):
        a = torch.from_numpy(np.asarray(action, dtype=np.float64).reshape(1, 2)).to(device())
        self._v.step(a)
        out = self._out.cpu().numpy()
        f = out[:64].view(np.float64)
        return f[:7].copy(), float(f[7]), bool(out[64])

    def reset(self):
        u = torch.from_numpy(np.random.rand(2).reshape(1, 2)).to(device())   # the global stream, like arm_env.py:66,69
        return self._v.reset(u).cpu().numpy()[0]

    def render(self):
        pass   # the pyglet viewer is outside the hot path

    def sample_action(self):
        return np.random.uniform(*self.action_bound, size=self.action_dim)

    def set_fps(self, fps=30):
        pass
