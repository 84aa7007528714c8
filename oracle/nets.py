"""Oracle (test infrastructure): the four MLPs of the reference, forward and backward.

NumPy restatement of
  * `Actor._build_net`          GAN-RL/DDPG.py:78-95
  * `Critic._build_net`         GAN-RL/DDPG.py:153-172
  * `Generator._build_net`      GAN-RL/Curious_GAN.py:112-135
  * `Discriminator._build_net`  GAN-RL/Curious_GAN.py:46-82
with hand-derived gradients that follow TensorFlow-1.x op semantics
(ReluGrad, Relu6Grad, TanhGrad, MaximumGrad tie rule `x >= y`).
Parity unpinned against TensorFlow itself (not installable here); an
independent torch.autograd derivation cross-checks it in the tests.

Parameters are kept as an ordered dict `name -> ndarray` in TensorFlow
variable-creation order; `flatten`/`unflatten` give the flat vector the
gradient exchange and the optimizers work on.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

STATE_DIM = 7   # arm_env.py:21
ACTION_DIM = 2  # arm_env.py:20
HIDDEN = 200    # DDPG.py:82,85,159 ; Curious_GAN.py:59,67,121,127


@dataclass(frozen=True)
class NetSpec:
    kind: str
    # first-layer input blocks in the order the reference ADDS the matmuls
    # (name of the input, name of the weight tensor, width)
    first: Tuple[Tuple[str, str, int], ...]
    # (tensor name, shape) in TF variable creation order
    tensors: Tuple[Tuple[str, Tuple[int, ...]], ...]
    widths: Tuple[int, int, int, int]
    acts: Tuple[str, str, str, str]
    bias_init: Tuple[float, float, float, float]

    @property
    def n_params(self) -> int:
        return int(sum(int(np.prod(s)) for _, s in self.tensors))

    @property
    def out_dim(self) -> int:
        return self.widths[3]


def actor_spec(s_dim=STATE_DIM, a_dim=ACTION_DIM) -> NetSpec:
    # DDPG.py:80-93: bias 0.001 on l1..l3, the 'a' layer has no bias_initializer -> zeros.
    return NetSpec(
        kind="actor",
        first=(("s", "l1/kernel", s_dim),),
        tensors=(("l1/kernel", (s_dim, HIDDEN)), ("l1/bias", (HIDDEN,)),
                 ("l2/kernel", (HIDDEN, HIDDEN)), ("l2/bias", (HIDDEN,)),
                 ("l3/kernel", (HIDDEN, 10)), ("l3/bias", (10,)),
                 ("a/a/kernel", (10, a_dim)), ("a/a/bias", (a_dim,))),
        widths=(HIDDEN, HIDDEN, 10, a_dim),
        acts=("relu6", "relu6", "relu", "tanh"),
        bias_init=(0.001, 0.001, 0.001, 0.0),
    )


def critic_spec(s_dim=STATE_DIM, a_dim=ACTION_DIM) -> NetSpec:
    # DDPG.py:155-171: every bias 0.01 (the q layer passes bias_initializer=init_b).
    return NetSpec(
        kind="critic",
        first=(("s", "l1/w1_s", s_dim), ("a", "l1/w1_a", a_dim)),
        tensors=(("l1/w1_s", (s_dim, HIDDEN)), ("l1/w1_a", (a_dim, HIDDEN)), ("l1/b1", (1, HIDDEN)),
                 ("l2/kernel", (HIDDEN, HIDDEN)), ("l2/bias", (HIDDEN,)),
                 ("l3/kernel", (HIDDEN, 10)), ("l3/bias", (10,)),
                 ("q/dense/kernel", (10, 1)), ("q/dense/bias", (1,))),
        widths=(HIDDEN, HIDDEN, 10, 1),
        acts=("relu6", "relu6", "relu", "linear"),
        bias_init=(0.01, 0.01, 0.01, 0.01),
    )


def generator_spec(s_dim=STATE_DIM, a_dim=ACTION_DIM) -> NetSpec:
    # Curious_GAN.py:114-134
    return NetSpec(
        kind="generator",
        first=(("s", "l1/w1_s", s_dim), ("a", "l1/w1_a", a_dim)),
        tensors=(("l1/w1_s", (s_dim, HIDDEN)), ("l1/w1_a", (a_dim, HIDDEN)), ("l1/b1", (1, HIDDEN)),
                 ("l2/kernel", (HIDDEN, HIDDEN)), ("l2/bias", (HIDDEN,)),
                 ("l3/kernel", (HIDDEN, 50)), ("l3/bias", (50,)),
                 ("l4/kernel", (50, s_dim)), ("l4/bias", (s_dim,))),
        widths=(HIDDEN, HIDDEN, 50, s_dim),
        acts=("relu", "relu", "relu", "linear"),
        bias_init=(0.001, 0.001, 0.001, 0.001),
    )


def discriminator_spec(s_dim=STATE_DIM, a_dim=ACTION_DIM) -> NetSpec:
    # Curious_GAN.py:52-81; creation order w1_G, w1_s, w1_a, b1 (:60-63),
    # summation order s, G, a, b1 (:64).  The output of the net is the LOGIT.
    return NetSpec(
        kind="discriminator",
        first=(("s", "l1/w1_s", s_dim), ("x", "l1/w1_G", s_dim), ("a", "l1/w1_a", a_dim)),
        tensors=(("l1/w1_G", (s_dim, HIDDEN)), ("l1/w1_s", (s_dim, HIDDEN)),
                 ("l1/w1_a", (a_dim, HIDDEN)), ("l1/b1", (1, HIDDEN)),
                 ("l2/kernel", (HIDDEN, HIDDEN)), ("l2/bias", (HIDDEN,)),
                 ("l3/kernel", (HIDDEN, 10)), ("l3/bias", (10,)),
                 ("D/w3", (10, 1)), ("D/b3", (1, 1))),
        widths=(HIDDEN, HIDDEN, 10, 1),
        acts=("lrelu", "lrelu", "lrelu", "linear"),
        bias_init=(0.001, 0.001, 0.001, 0.001),
    )


SPECS = {"actor": actor_spec, "critic": critic_spec,
         "generator": generator_spec, "discriminator": discriminator_spec}


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
def _is_bias(name: str) -> bool:
    leaf = name.split("/")[-1]
    return leaf in ("bias", "b1", "b3")


def _layer_of(spec: NetSpec, name: str) -> int:
    head = name.split("/")[0]
    return {"l1": 0, "l2": 1, "l3": 2}.get(head, 3)


def init_params(spec: NetSpec, rng: np.random.RandomState, dtype=np.float64) -> "OrderedDict[str, np.ndarray]":
    """xavier-uniform weights (limit sqrt(6/(fan_in+fan_out)), tf.contrib.layers.xavier_initializer
    defaults, DDPG.py:80) drawn tensor by tensor in creation order; constant biases."""
    p = OrderedDict()
    for name, shape in spec.tensors:
        if _is_bias(name):
            p[name] = np.full(shape, spec.bias_init[_layer_of(spec, name)], dtype=dtype)
        else:
            limit = np.sqrt(6.0 / (shape[0] + shape[1]))
            p[name] = rng.uniform(-limit, limit, size=shape).astype(dtype)
    return p


def flatten(spec: NetSpec, params: Dict[str, np.ndarray]) -> np.ndarray:
    return np.concatenate([np.asarray(params[n]).reshape(-1) for n, _ in spec.tensors])


def unflatten(spec: NetSpec, flat: np.ndarray) -> "OrderedDict[str, np.ndarray]":
    out, o = OrderedDict(), 0
    for n, shape in spec.tensors:
        k = int(np.prod(shape))
        out[n] = np.asarray(flat[o:o + k]).reshape(shape).copy()
        o += k
    assert o == flat.size
    return out


def _tail_names(spec: NetSpec):
    """(kernel, bias) tensor names of layers 2..4."""
    names = [n for n, _ in spec.tensors]
    first_w = {w for _, w, _ in spec.first}
    rest = [n for n in names if n not in first_w and not n.startswith("l1/")]
    assert len(rest) == 6, rest
    return [(rest[0], rest[1]), (rest[2], rest[3]), (rest[4], rest[5])]


def _first_bias(spec: NetSpec) -> str:
    return [n for n, _ in spec.tensors if n.startswith("l1/") and _is_bias(n)][0]


# --------------------------------------------------------------------------
# activations (forward on z, derivative mask from z)
# --------------------------------------------------------------------------
LRELU_ALPHA = 0.2  # Curious_GAN.py:56


def act_fwd(kind: str, z: np.ndarray) -> np.ndarray:
    if kind == "relu":
        return np.maximum(z, 0)
    if kind == "relu6":
        return np.minimum(np.maximum(z, 0), 6)
    if kind == "lrelu":      # tf.maximum(alpha*net, net)  Curious_GAN.py:65
        return np.maximum(z.dtype.type(LRELU_ALPHA) * z, z)
    if kind == "tanh":
        return np.tanh(z)
    if kind == "linear":
        return z
    raise ValueError(kind)


def act_bwd(kind: str, z: np.ndarray, y: np.ndarray, dy: np.ndarray) -> np.ndarray:
    """dL/dz from dL/dy.  relu: z>0; relu6: 0<z<6 (Relu6Grad); lrelu: MaximumGrad routes the
    gradient to the `alpha*net` branch when alpha*net >= net, i.e. z <= 0 (tie at 0 -> alpha);
    tanh: TanhGrad uses the OUTPUT y."""
    t = z.dtype.type
    if kind == "relu":
        return dy * (z > 0)
    if kind == "relu6":
        return dy * ((z > 0) & (z < 6))
    if kind == "lrelu":
        return dy * np.where(z > 0, t(1), t(LRELU_ALPHA))
    if kind == "tanh":
        return dy * (t(1) - y * y)
    if kind == "linear":
        return dy
    raise ValueError(kind)


# --------------------------------------------------------------------------
# forward / backward
# --------------------------------------------------------------------------
def forward(spec: NetSpec, params: Dict[str, np.ndarray], inputs: Dict[str, np.ndarray]):
    """Returns (output, cache).  `inputs` maps the first-layer input names of the spec
    ('s', 'a', 'x') to [B, width] arrays already in the compute dtype."""
    z = None
    for inp, wname, _ in spec.first:
        term = inputs[inp] @ params[wname]
        z = term if z is None else z + term
    z = z + params[_first_bias(spec)].reshape(1, -1)
    zs, hs = [z], [act_fwd(spec.acts[0], z)]
    for li, (wn, bn) in enumerate(_tail_names(spec), start=1):
        z = hs[-1] @ params[wn] + params[bn].reshape(1, -1)
        zs.append(z)
        hs.append(act_fwd(spec.acts[li], z))
    cache = {"inputs": inputs, "z": zs, "h": hs}
    return hs[-1], cache


def backward(spec: NetSpec, params: Dict[str, np.ndarray], cache, d_out: np.ndarray,
             out_override: np.ndarray | None = None):
    """Back-propagate `d_out` (dL/d output, [B, out]) through the net.

    Returns (grads, d_inputs): `grads` has one entry per tensor of the spec with the
    batch-SUMMED gradient (callers put any 1/B into `d_out`), `d_inputs` the gradient
    with respect to each first-layer input block.

    `out_override`: value to use as the OUTPUT of the last activation in its gradient
    (TanhGrad reads the op's output tensor; the reference feeds that tensor in
    `Critic.learn`, DDPG.py:175, so the fed action replaces tanh's own output there).
    """
    zs, hs, inputs = cache["z"], cache["h"], cache["inputs"]
    grads = OrderedDict((n, None) for n, _ in spec.tensors)
    tails = _tail_names(spec)
    y_last = hs[3] if out_override is None else out_override
    dz = act_bwd(spec.acts[3], zs[3], y_last, d_out)
    for li in (3, 2, 1):
        wn, bn = tails[li - 1]
        grads[wn] = hs[li - 1].T @ dz
        grads[bn] = dz.sum(axis=0).reshape(params[bn].shape)
        dh = dz @ params[wn].T
        dz = act_bwd(spec.acts[li - 1], zs[li - 1], hs[li - 1], dh)
    b1 = _first_bias(spec)
    grads[b1] = dz.sum(axis=0).reshape(params[b1].shape)
    d_inputs = {}
    for inp, wname, _ in spec.first:
        grads[wname] = inputs[inp].T @ dz
        d_inputs[inp] = dz @ params[wname].T
    return grads, d_inputs


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def sigmoid(x):
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1 / (1 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1 + e)
    return out


def sigmoid_ce(logits: np.ndarray, labels) -> np.ndarray:
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))
    (Curious_GAN.py:33-39,139)."""
    t = logits.dtype.type
    return np.maximum(logits, 0) - logits * t(labels) + np.log1p(np.exp(-np.abs(logits)))
