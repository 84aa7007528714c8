"""CPU-side helpers shared by the parity tests and the fixture generator: an oracle agent with
weights that exercise every activation branch, and synthetic replay batches."""
import numpy as np

from oracle import nets as onets
from oracle.agent import CuriousAgent, Hyper

NETS = ("actor_e", "actor_t", "critic_e", "critic_t", "gen", "disc")


def _param_sets(o):
    return (o.actor_e, o.actor_t, o.critic_e, o.critic_t, o.gen, o.disc)


def round_to_f32(o):
    """The device holds fp32 weights: start the float64 oracle from the same representable values."""
    for p in _param_sets(o):
        for k in p:
            p[k] = p[k].astype(np.float32).astype(np.float64)


def make_oracle(seed=1234, hyper=None, spread=True):
    o = CuriousAgent(seed=seed, dtype=np.float64, hyper=hyper or Hyper())
    if spread:
        # make clipped / negative-side activations common so every mask branch is exercised
        rng = np.random.RandomState(seed + 1)
        for p in _param_sets(o):
            for k in p:
                if k.startswith("l1/") and p[k].shape[0] > 1:
                    p[k] = p[k] * 6.0
                if "bias" in k or k.endswith("b1") or k.endswith("b3"):
                    p[k] = p[k] + rng.uniform(-0.3, 0.3, p[k].shape)
    round_to_f32(o)
    return o


def load_weights(o, flats):
    """Set the six networks of an oracle agent from flat vectors (dict name -> array)."""
    spec = {"actor_e": o.actor_spec, "actor_t": o.actor_spec, "critic_e": o.critic_spec,
            "critic_t": o.critic_spec, "gen": o.gen_spec, "disc": o.disc_spec}
    for k in NETS:
        setattr(o, k, onets.unflatten(spec[k], np.asarray(flats[k], dtype=np.float64)))


def make_batch(B, seed=0, f32=True):
    """[B,25] rows in the 6-field layout with plausible ranges (states ~[-1.5,1.5], actions [-1,1])."""
    rng = np.random.RandomState(seed)
    b = np.empty((B, 25))
    b[:, 0:21] = rng.uniform(-1.5, 1.5, (B, 21))
    b[:, 0] = rng.randint(0, 2, B)
    b[:, 21:23] = rng.uniform(-1, 1, (B, 2))
    b[:, 23] = np.where(rng.rand(B) < 0.1, 10.0, 0.0)
    b[:, 24] = b[:, 23] + 0.01 * rng.rand(B)
    if f32:
        b = b.astype(np.float32).astype(np.float64)
    return b
