"""Inert stand-ins for the TensorFlow objects the reference's constructors take.

The reference wires its classes together with a `tf.Session`, placeholders `S, S_, R, G` and the
output tensors `actor.a`, `actor.a_`, `generator.G`, `critic.a_grads`, `discriminator.D_logit_fake`
(Curious_GAN.py:346-362).  Here a placeholder only carries its shape and a tensor handle only
names its producer, so the same constructor calls keep working while all state lives on the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional, Sequence

import numpy as np

_INIT_RNG = np.random.RandomState()


def set_random_seed(seed: int) -> None:
    """Seed of the weight initialisers, the counterpart of tf.set_random_seed (Curious_GAN.py:321).
    Networks draw their xavier weights from this stream in construction order."""
    global _INIT_RNG
    _INIT_RNG = np.random.RandomState(seed)


def init_rng() -> np.random.RandomState:
    return _INIT_RNG


class Session:
    """Accepted wherever the reference takes `sess`; holds the engine shared by the objects built
    with it so work arrays are reused."""

    def __init__(self):
        self._engine = None

    def engine(self, s_dim: int, a_dim: int):
        from .engine import StepEngine
        if self._engine is None:
            self._engine = StepEngine(s_dim, a_dim)
        assert (self._engine.s_dim, self._engine.a_dim) == (s_dim, a_dim)
        return self._engine

    def run(self, *a, **k):
        raise NotImplementedError("there is no graph to run: call the objects' methods")

    def close(self):
        pass


@dataclass
class Placeholder:
    shape: Sequence[Optional[int]]
    name: str = ""

    @property
    def width(self) -> int:
        return int(self.shape[-1])


def placeholder(dtype: Any = None, shape: Sequence[Optional[int]] = (None, 1), name: str = "") -> Placeholder:
    """tf.placeholder(tf.float32, shape=[None, STATE_DIM], name='s') (Curious_GAN.py:346-353)."""
    return Placeholder(tuple(shape), name)


@dataclass
class TensorHandle:
    owner: Any
    role: str
