"""Diagnostic: how close each gradient tensor of the four steps comes to the parity tolerance (max of err / tol)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gpu_util import Pair, dev, make_batch, make_smooth_batch, RTOL

def ratio(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    tol = RTOL * np.abs(ref) + RTOL * np.abs(ref).max()
    r = np.abs(got - ref) / tol
    return float(r.max()), int((r > 1).sum())

for B in [int(a) for a in sys.argv[1:]] or [4096, 32768]:
    p = Pair(); b = (make_smooth_batch(p.o, B, seed=B)[0] if os.environ.get('SMOOTH') else make_batch(B, seed=B)); bt = dev(b)
    f = dict(s=b[:, :7], s_=b[:, 7:14], g=b[:, 14:21], a=b[:, 21:23], cr=b[:, 24:25])
    ft = dict(s=bt[:, :7], s_=bt[:, 7:14], g=bt[:, 14:21], a=bt[:, 21:23], cr=bt[:, 24])
    ref = p.o.critic_learn(f["s"], f["a"], f["cr"], f["s_"])
    wc, wa = p.eng.critic_step(p.C, p.A, ft["s"], ft["a"], ft["cr"], ft["s_"])
    print(B, "critic", ratio(wc.grad.cpu().numpy(), ref["g_critic"]), "actor-through-critic", ratio(wa.grad.cpu().numpy(), ref["g_actor"]),
          "loss", abs(p.eng.losses[0].item() - ref["loss"]) / ref["loss"])
    ref = p.o.actor_learn(f["s"]); wa2 = p.eng.actor_step(p.A, p.C, ft["s"])
    print(B, "actor", ratio(wa2.grad.cpu().numpy(), ref["g_actor"]))
    ref = p.o.disc_learn(f["g"], f["s_"], f["s"], f["a"]); wd = p.eng.disc_step(p.D, ft["g"], ft["s_"], ft["s"], ft["a"])
    print(B, "disc", ratio(wd.grad.cpu().numpy(), ref["g_disc"]))
    ref = p.o.gen_learn(f["s"], f["a"]); wg = p.eng.gen_step(p.G, p.D, ft["s"], ft["a"])
    print(B, "gen", ratio(wg.grad.cpu().numpy(), ref["g_gen"]))
