"""CPU oracle for the RL-GAN hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference's arithmetic
(`GAN-RL/DDPG.py`, `GAN-RL/Curious_GAN.py`, `GAN-RL/arm_env.py`) used as the
checker for the HIP kernels.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product package never
does, and the product path fails loudly when its HIP library is missing.

Pinning status
--------------
* `arm_env`, `memory`, `mt19937` (the NumPy half of the reference): pinned by
  golden vectors produced by importing the reference modules in the build
  container (`tests/golden/gen_fixtures.py`) and by NumPy's own
  `RandomState` stream.
* `nets`, `optim`, `agent` (the TensorFlow-1.x half): **parity unpinned** --
  TensorFlow is not installable in the build container and the reference ships
  no tests, fixtures or checkpoints for this path.  The restatement follows
  the documented TF-1.x op semantics, cites the reference call site of every
  op, and is cross-checked against an independent derivation with
  `torch.autograd` in `tests/test_oracle_autograd.py`.
"""
