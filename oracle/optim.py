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
    """Besides the update rule this class keeps, per entry, a bound `allow` on how far the variable may end up from
    the value computed here when every gradient handed to `step` is replaced by one within the tolerance gradients
    are held to, |dg| <= grad_rtol * (|g| + max|g|) (and dg = 0 where g is exactly 0).  Adam divides by sqrt(v), so where a gradient entry is itself
    rounding noise (|g| far below max|g|) the step is about lr in a direction the noise decides, while an entry with
    a healthy gradient moves by lr * grad_rtol * (1 + max|g| / |g|), next to nothing.  The bound follows the error
    through the two slots: dm <- beta1*dm + (1-beta1)*dg bounds the error of m, dv likewise that of v (through
    2|g|dg), and a step's share is lr_t * (dm / (sqrt(v)+eps) + |m| * dv / (2 sqrt(v) (sqrt(v)+eps)^2)), capped at
    2*lr (a step cannot be more wrong than a full step the other way).  Tests use it as the absolute tolerance of
    Adam-trained weights; it is not part of the update."""

    def __init__(self, n: int, lr: float, dtype=np.float64, beta1=0.9, beta2=0.999, eps=1e-8, grad_rtol=1e-5):
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.m = np.zeros(n, dtype=dtype)
        self.v = np.zeros(n, dtype=dtype)
        self.t = 0
        self.grad_rtol = grad_rtol
        self.allow = np.zeros(n)
        self._dm = np.zeros(n)
        self._dv = np.zeros(n)

    def lr_t(self, t: int) -> float:
        return self.lr * np.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)

    def step(self, var: np.ndarray, g: np.ndarray) -> np.ndarray:
        ty = var.dtype.type
        self.t += 1
        self.m = ty(self.beta1) * self.m + ty(1 - self.beta1) * g
        self.v = ty(self.beta2) * self.v + ty(1 - self.beta2) * g * g
        self._track(np.asarray(g, dtype=np.float64))
        return var - ty(self.lr_t(self.t)) * self.m / (np.sqrt(self.v) + ty(self.eps))

    def _track(self, g: np.ndarray) -> None:
        ag = np.abs(g)
        # an entry that is exactly zero here is zero term by term (inactive units), which float32 reproduces exactly
        dg = self.grad_rtol * (ag + (ag.max() if ag.size else 0.0)) * (ag > 0)
        self._dm = self.beta1 * self._dm + (1 - self.beta1) * dg
        self._dv = self.beta2 * self._dv + (1 - self.beta2) * 2.0 * ag * dg
        rv = np.sqrt(np.asarray(self.v, dtype=np.float64))
        den = rv + self.eps
        share = self._dm / den + np.abs(np.asarray(self.m, dtype=np.float64)) * self._dv / (2.0 * np.maximum(rv, 1e-300) * den * den)
        self.allow = self.allow + np.minimum(2.0 * abs(self.lr), self.lr_t(self.t) * share)
