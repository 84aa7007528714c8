"""Oracle (test infrastructure): the 2-link arm environment, scalar and batched.

Restates `ArmEnv.step/reset/_get_state/_r_func` (GAN-RL/arm_env.py:43-78,91-115) in
float64 NumPy.  Pinned: `tests/golden/arm_env_*.npz` hold trajectories produced by the
reference module itself (see `tests/golden/gen_fixtures.py`).

Behaviours kept on purpose:
  * the goal array is integer-typed (arm_env.py:39), so a hard-mode reset TRUNCATES the
    random goal toward zero (arm_env.py:66-67);
  * a hard-mode reset neither resets the joint angles nor recomputes the link ends
    (arm_env.py:65-67): at the very first reset the link ends are still (0, 0);
  * `%=` is Python's sign-of-divisor modulo: the angle stays in [0, 2*pi) (arm_env.py:47);
  * the `in_point` flag of the state is read BEFORE `_r_func` updates the counter
    (arm_env.py:56-57), so it lags one step.
"""
from __future__ import annotations

import numpy as np

DT = 0.1
ARM_L = 100.0
VIEW = 400.0
POINT_L = 15.0
GRAB_T = 50
TWO_PI = np.pi * 2


class ScalarArmEnv:
    """One environment, same call sequence as the reference class."""
    state_dim, action_dim, action_bound = 7, 2, [-1, 1]

    def __init__(self, mode="easy", rng=None):
        self.mode = mode
        self.rng = rng if rng is not None else np.random  # reference draws from the global stream
        self.theta = np.zeros(2)
        self.ends = np.zeros((2, 2))           # (x1,y1),(x2,y2)  arm_env.py:36
        self.goal = np.array([250, 303])       # int array, arm_env.py:39
        self.goal_init = self.goal.copy()
        self.center = np.array([VIEW, VIEW]) / 2
        self.grab_counter = 0
        self.get_point = False

    def _ends(self):
        d1 = np.array([ARM_L * np.cos(self.theta[0]), ARM_L * np.sin(self.theta[0])])
        d2 = np.array([ARM_L * np.cos(self.theta[1]), ARM_L * np.sin(self.theta[1])])
        self.ends[0] = self.center + d1
        self.ends[1] = self.ends[0] + d2

    def _state(self):
        t_arms = np.ravel(self.ends - self.goal)
        center_dis = (self.center - self.goal) / 200
        in_point = 1 if self.grab_counter > 0 else 0
        return np.hstack([in_point, t_arms / 200, center_dis]), t_arms[-2:]

    def step(self, action):
        action = np.clip(action, -1, 1)
        self.theta += action * DT
        self.theta %= TWO_PI
        self._ends()
        s, d = self._state()
        dist = np.sqrt(np.sum(np.square(d)))
        r = 0.0
        if dist < POINT_L and not self.get_point:
            self.grab_counter += 1
            if self.grab_counter > GRAB_T:
                r += 10.0
                self.get_point = True
        elif dist > POINT_L:
            self.grab_counter = 0
            self.get_point = False
        return s, r, self.get_point

    def reset(self):
        self.get_point = False
        self.grab_counter = 0
        if self.mode == "hard":
            pxy = np.clip(self.rng.rand(2) * VIEW, 100, 300)
            self.goal[:] = pxy                  # truncating assignment into the int array
        else:
            self.theta[:] = self.rng.rand(2) * TWO_PI
            self._ends()
            self.goal[:] = self.goal_init
        return self._state()[0]


class VecArmEnv:
    """N environments in lock-step (struct of arrays, float64), same arithmetic per env."""

    def __init__(self, n: int, mode="hard"):
        self.n, self.mode = n, mode
        self.theta = np.zeros((n, 2))
        self.ends = np.zeros((n, 4))            # x1,y1,x2,y2
        self.goal = np.tile(np.array([250.0, 303.0]), (n, 1))   # integer-VALUED floats
        self.grab = np.zeros(n, dtype=np.int32)
        self.got = np.zeros(n, dtype=bool)

    def state(self):
        s = np.empty((self.n, 7))
        s[:, 0] = (self.grab > 0)
        s[:, 1] = (self.ends[:, 0] - self.goal[:, 0]) / 200
        s[:, 2] = (self.ends[:, 1] - self.goal[:, 1]) / 200
        s[:, 3] = (self.ends[:, 2] - self.goal[:, 0]) / 200
        s[:, 4] = (self.ends[:, 3] - self.goal[:, 1]) / 200
        s[:, 5] = (200.0 - self.goal[:, 0]) / 200
        s[:, 6] = (200.0 - self.goal[:, 1]) / 200
        return s

    def _recompute_ends(self, idx=slice(None)):
        th = self.theta[idx]
        x1 = 200.0 + ARM_L * np.cos(th[:, 0])
        y1 = 200.0 + ARM_L * np.sin(th[:, 0])
        self.ends[idx, 0], self.ends[idx, 1] = x1, y1
        self.ends[idx, 2] = x1 + ARM_L * np.cos(th[:, 1])
        self.ends[idx, 3] = y1 + ARM_L * np.sin(th[:, 1])

    def step(self, action):
        a = np.clip(np.asarray(action, dtype=np.float64), -1, 1)
        self.theta = np.mod(self.theta + a * DT, TWO_PI)
        self._recompute_ends()
        s = self.state()
        dx = self.ends[:, 2] - self.goal[:, 0]
        dy = self.ends[:, 3] - self.goal[:, 1]
        dist = np.sqrt(dx * dx + dy * dy)
        r = np.zeros(self.n)
        near = (dist < POINT_L) & (~self.got)
        far = (~near) & (dist > POINT_L)
        self.grab = np.where(near, self.grab + 1, self.grab)
        hit = near & (self.grab > GRAB_T)
        r[hit] = 10.0
        self.got = np.where(hit, True, self.got)
        self.grab = np.where(far, 0, self.grab)
        self.got = np.where(far, False, self.got)
        return s, r, self.got.copy()

    def reset(self, mask, u):
        """Reset the envs selected by `mask`; `u` is [n,2] uniform [0,1) draws (only the masked
        rows are consumed)."""
        mask = np.asarray(mask, dtype=bool)
        self.got[mask] = False
        self.grab[mask] = 0
        if self.mode == "hard":
            pxy = np.clip(u * VIEW, 100, 300)
            self.goal[mask] = np.trunc(pxy[mask])
        else:
            self.theta[mask] = u[mask] * TWO_PI
            idx = np.nonzero(mask)[0]
            if idx.size:
                self._recompute_ends(idx)
            self.goal[mask] = np.array([250.0, 303.0])
        return self.state()
