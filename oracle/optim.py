"""Oracle (test infrastructure): TensorFlow-1.x optimizer update rules on flat vectors.

* `tf.train.RMSPropOptimizer(lr)` (DDPG.py:112,148): defaults decay=0.9, momentum=0,
  epsilon=1e-10; the `rms` slot starts at ONE, epsilon sits INSIDE the square root:
      ms  <- decay*ms + (1-decay)*g*g
      mom <- momentum*mom + lr*g/sqrt(ms + eps)
      var <- var - mom
* `tf.train.AdamOptimizer(lr)` (Curious_GAN.py:45,142): beta1=0.9, beta2=0.999, eps=1e-8
  OUTSIDE the square root, with the bias correction folded into the step size:
      lr_t <- lr*sqrt(1-beta2^t)/(1-beta1^t)      (t = 1 on the first step)
      m <- beta1*m + (1-beta1)*g ; v <- beta2*v + (1-beta2)*g*g
      var <- var - lr_t*m/(sqrt(v)+eps)
Parity unpinned against TensorFlow (not installable here).
"""
from __future__ import annotations

import numpy as np


class RMSProp:
    def __init__(self, n: int, lr: float, dtype=np.float64, decay=0.9, momentum=0.0, eps=1e-10):
        self.lr, self.decay, self.momentum, self.eps = lr, decay, momentum, eps
        self.ms = np.ones(n, dtype=dtype)
        self.mom = np.zeros(n, dtype=dtype)

    def step(self, var: np.ndarray, g: np.ndarray) -> np.ndarray:
        t = var.dtype.type
        self.ms = t(self.decay) * self.ms + t(1 - self.decay) * g * g
        self.mom = t(self.momentum) * self.mom + t(self.lr) * g / np.sqrt(self.ms + t(self.eps))
        return var - self.mom


class Adam:
    def __init__(self, n: int, lr: float, dtype=np.float64, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.m = np.zeros(n, dtype=dtype)
        self.v = np.zeros(n, dtype=dtype)
        self.t = 0

    def lr_t(self, t: int) -> float:
        return self.lr * np.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)

    def step(self, var: np.ndarray, g: np.ndarray) -> np.ndarray:
        ty = var.dtype.type
        self.t += 1
        self.m = ty(self.beta1) * self.m + ty(1 - self.beta1) * g
        self.v = ty(self.beta2) * self.v + ty(1 - self.beta2) * g * g
        return var - ty(self.lr_t(self.t)) * self.m / (np.sqrt(self.v) + ty(self.eps))
