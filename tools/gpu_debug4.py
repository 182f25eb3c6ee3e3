import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import torch, sship
m, n = 2048, 16384
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32) / np.sqrt(m)
h = sship.Homotopy(A)
def batch(ks):
    rng = np.random.default_rng(3)
    Y = torch.empty((len(ks), m), device=dev)
    for b, k in enumerate(ks):
        sup = np.sort(rng.choice(n, k, replace=False)); coef = 1 + np.abs(rng.standard_normal(k))
        Y[b] = (A[:, torch.from_numpy(sup).to(dev)].double() @ torch.from_numpy(coef).to(dev)).float()
    return Y
for name, ks in (("all k=40", [40] * 256), ("tile0 k=1, tile1 k=40", [1] * 128 + [40] * 128), ("1 hard of 256", [1] * 255 + [40])):
    Y = batch(ks)
    X = torch.zeros((256, n), device=dev)
    h.solve_batch(Y, 1e-3, 100, out=X)
    torch.cuda.synchronize(); h.reset_stats(); t0 = time.perf_counter()
    _, it, _ = h.solve_batch(Y, 1e-3, 100, out=X)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-24s %.1f ms rounds %d iters %d..%d" % (name, dt * 1e3, h.stats()["batch_rounds"], it.min(), it.max()))
