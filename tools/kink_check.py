"""Diagnostic (CPU): pre-activations of the critic's forward passes at batch B that lie within float32 rounding of a kink of
relu / relu6 (0 or 6), where the derivative mask depends on the last bit of a float32 sum."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import nets
from oracle_util import make_batch, make_oracle

for B in [int(a) for a in sys.argv[1:]]:
    o = make_oracle(1234)
    b = make_batch(B, seed=B)
    s, a = b[:, :7], b[:, 21:23]
    for name, spec, params, inp in (("critic_e(s,a)", o.critic_spec, o.critic_e, {"s": s, "a": a}), ("actor_e(s)", o.actor_spec, o.actor_e, {"s": s})):
        _, cache = nets.forward(spec, params, inp)
        for li, z in enumerate(cache["z"][:3]):
            scale = np.abs(z).max()
            for kink in (0.0, 6.0):
                d = np.abs(z - kink)
                near = np.argwhere(d < 4e-7 * scale)
                if near.size:
                    print(B, name, "layer", li + 1, "kink", kink, "entries within 4e-7*scale:", [(int(r), int(c), float(z[r, c] - kink)) for r, c in near[:6]])
    print(B, "done")
