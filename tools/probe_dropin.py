"""Developer probe: the drop-in surface (sparsesolvers.Homotopy, host arrays) next to a sship context in one process."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship, sparsesolvers
M, N, K = 8192, 65536, 64
A = np.random.default_rng(1234).standard_normal((M, N), dtype=np.float32)
A /= np.float32(np.sqrt(M))
ys = []
for s in range(10):
    rng = np.random.default_rng(1235 + s)
    sup = np.sort(rng.choice(N, K, replace=False))
    coef = 1.0 + np.abs(rng.standard_normal(K))
    ys.append((A[:, sup].astype(np.float64) @ coef).astype(np.float32))
Ad = torch.from_numpy(A).to("cuda:0")
hs = [sship.Homotopy(Ad) for _ in range(3)]
solvers = [sparsesolvers.Homotopy(A) for _ in range(4)]
for i, s in enumerate(solvers):
    s.solve(ys[0], tolerance=1e-3, max_iterations=256)
    t0 = time.perf_counter()
    for y in ys[1:]:
        s.solve(y, tolerance=1e-3, max_iterations=256)
    print("drop-in context %d: %.3f ms per solve (host arrays)" % (i, (time.perf_counter() - t0) / 9 * 1e3))
