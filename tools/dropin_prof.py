import os, sys, time, collections
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rl_gan_amd as R
cfg = R.Settings(MEMORY_CAPACITY=400, BATCH_SIZE=16)
ag = R.Agent(cfg)
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[type(obj).__name__ + "." + name] += time.perf_counter() - t; cnt[type(obj).__name__ + "." + name] += 1; return r
    setattr(obj, name, g)
R.train(ag, episodes=3, log=None)
for o, ns in ((ag.actor, ["choose_action", "learn"]), (ag.critic, ["learn"]), (ag.generator, ["predict", "learn", "eval"]),
              (ag.discriminator, ["determine", "learn", "eval"]), (ag.M, ["store_transition", "sample"]), (ag.env, ["step", "reset"])):
    for n in ns: wrap(o, n)
t0 = time.perf_counter(); p0 = ag.M.pointer
R.train(ag, episodes=5, log=None)
torch.cuda.synchronize(); dt = time.perf_counter() - t0; n = ag.M.pointer - p0
print(f"{n} transitions, {dt / n * 1e6:.0f} us each")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:32s} {v / cnt[k] * 1e6:8.1f} us/call  x{cnt[k] / n:.2f}/transition = {v / n * 1e6:7.1f} us")
