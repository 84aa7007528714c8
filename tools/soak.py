import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rl_gan_amd as R
L = R.CuriousLearner(R.LearnerConfig(n_env=4096, batch_size=4096, memory_capacity=1 << 20, seed=1))
L.fill(); L.capture()
t0 = time.perf_counter()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30000):
    L.step()
    if i % 10000 == 9999:
        torch.cuda.synchronize()
        l = L.losses()
        print(i + 1, "steps", {k: round(v, 5) for k, v in l.items()}, f"{(i + 1) / (time.perf_counter() - t0):.0f} updates/s", flush=True)
        assert all(np.isfinite(v) for v in l.values())
w = L.actor.e.get_flat()
assert np.isfinite(w).all()
print("actor |w| max", float(np.abs(w).max()), "resets", int(L.n_resets.item()), "pointer", L.memory.pointer)
