import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import rl_gan_amd as R
from rl_gan_amd._lib import lib, check
from rl_gan_amd.core import _stream
rng = np.random.RandomState(0)
for m, k, n in [(4096, 264, 200), (300, 264, 200), (4096, 200, 200)]:
    x = rng.uniform(0, 2, (m, k)).astype(np.float32)
    dz = (rng.standard_normal((m, n)) * 1e-3).astype(np.float32)
    L = R.DenseLayer(k, n, rng=rng)
    xt, dzt = torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda()
    y = torch.zeros(m, n, device="cuda")
    L.backward(xt, y, dzt, need_dx=False, act=False)
    got = L.grad.cpu().numpy()[:k * n].reshape(k, n).astype(np.float64)
    ref = x.astype(np.float64).T @ dz.astype(np.float64)
    tm = (xt.T @ dzt).cpu().numpy().astype(np.float64)
    sc = np.abs(ref).max()
    print(m, k, n, "ours max err/scale", np.abs(got - ref).max() / sc, " torch fp32 matmul", np.abs(tm - ref).max() / sc)
    # forward
    w = L.canon[:k*n].reshape(k, n)
    yf = L.forward(xt, act=False).cpu().numpy().astype(np.float64)
    rf = x.astype(np.float64) @ w.cpu().numpy().astype(np.float64) + L.canon[k*n:].cpu().numpy().astype(np.float64)
    print("   forward err/scale", np.abs(yf - rf).max() / np.abs(rf).max())
print("small-n products")
for m, k, n in [(4096, 200, 12), (304, 200, 12), (4096, 12, 4), (8192, 200, 12)]:
    x = rng.uniform(0, 2, (m, k)).astype(np.float32)
    dz = (rng.standard_normal((m, n)) * 1e-3).astype(np.float32)
    L = R.DenseLayer(k, n, rng=rng)
    xt, dzt = torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda()
    y = torch.zeros(m, n, device="cuda")
    dx = L.backward(xt, y, dzt, need_dx=True, act=False).cpu().numpy().astype(np.float64)
    w = L.canon[:k*n].reshape(k, n).cpu().numpy().astype(np.float64)
    rdx = dz.astype(np.float64) @ w.T
    got = L.grad.cpu().numpy()[:k * n].reshape(k, n).astype(np.float64)
    ref = x.astype(np.float64).T @ dz.astype(np.float64)
    bad_rows = np.where(np.abs(dx - rdx).max(1) > 1e-5 * np.abs(rdx).max())[0]
    print(m, k, n, "dW err/scale", np.abs(got - ref).max() / np.abs(ref).max(), " dX err/scale", np.abs(dx - rdx).max() / np.abs(rdx).max(),
          "bad rows", bad_rows[:10], len(bad_rows))
