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


# ---- minibatches without rows that sit on a kink -----------------------------------------------------------
KINK_DELTA = 1e-6      # relative to the layer's largest |pre-activation|: ten float32 roundings of a 200-term sum


def _near_kink_rows(spec, params, inputs, delta=KINK_DELTA):
    """Rows with a pre-activation within delta * max|z| of a point where the layer's activation has a corner."""
    _, cache = onets.forward(spec, params, inputs)
    bad = np.zeros(cache["z"][0].shape[0], dtype=bool)
    for kind, z in zip(spec.acts[:3], cache["z"][:3]):
        lim = delta * np.abs(z).max()
        if kind in ("relu", "relu6", "lrelu"):
            bad |= (np.abs(z) < lim).any(axis=1)
        if kind == "relu6":
            bad |= (np.abs(z - 6.0) < lim).any(axis=1)
    return bad


def kink_rows(o, batch, delta=KINK_DELTA):
    """Rows of a [B, 25] minibatch at which a backward pass of one of the four update steps (Curious_GAN.py:208-211),
    each taken from the weights `o` holds now, meets a pre-activation within rounding of a corner of relu / relu6 /
    leaky relu.  There the derivative is decided by the last bit of a float32 sum, so two correct float32
    implementations (TensorFlow's included) may disagree by that row's whole contribution to the gradient, and a
    float64 oracle cannot arbitrate.  (About 1.4 % of random rows; at batch 32,768 some row is on a corner in every
    draw, and which ones trip depends on the order of additions inside the kernels.)"""
    s, s_, g, a = batch[:, :7], batch[:, 7:14], batch[:, 14:21], batch[:, 21:23]
    near = lambda spec, params, inputs: _near_kink_rows(spec, params, inputs, delta)
    bad = near(o.critic_spec, o.critic_e, {"s": s, "a": a})                        # critic.learn
    bad |= near(o.actor_spec, o.actor_e, {"s": s})                                  # ... the actor through it; actor.learn
    bad |= near(o.critic_spec, o.critic_e, {"s": s, "a": o.choose_action(s)})       # actor.learn: live action
    bad |= near(o.disc_spec, o.disc, {"s": s, "x": g, "a": a})                      # discriminator.learn: fake, real
    bad |= near(o.disc_spec, o.disc, {"s": s, "x": s_, "a": a})
    bad |= near(o.gen_spec, o.gen, {"s": s, "a": a})                                # generator.learn through D
    bad |= near(o.disc_spec, o.disc, {"s": s, "x": o.predict(s, a), "a": a})
    return bad


def make_smooth_batch(o, B, seed=0, max_rounds=12):
    """make_batch(B, seed) with the rows of `kink_rows` (for the weights `o` holds now) drawn again until none is left.
    Returns the batch and the number of rows that were replaced."""
    b = make_batch(B, seed)
    replaced = 0
    for k in range(max_rounds):
        bad = kink_rows(o, b)
        n = int(bad.sum())
        if n == 0:
            return b, replaced
        b[bad] = make_batch(n, seed=seed * 1000 + 17 + k)
        replaced += n
    raise AssertionError("rows on a kink kept turning up")
