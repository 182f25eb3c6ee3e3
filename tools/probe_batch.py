#!/usr/bin/env python3
"""GPU probe: BASELINE configs[2]: B signals sharing the C2 matrix, GEMM form vs Gram form of the lock-step batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship, torch
m, n, k = 8192, 65536, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1234)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32) / np.sqrt(m)
rng = np.random.default_rng(7)
Y = torch.empty((B, m), device=dev, dtype=torch.float32)
sups = []
for b in range(B):
    sup = np.sort(rng.choice(n, k, replace=False))
    coef = torch.from_numpy((1.0 + np.abs(rng.standard_normal(k))).astype(np.float32)).to(dev)
    Y[b] = A[:, torch.from_numpy(sup).to(dev)] @ coef
    sups.append(sup)
X = torch.zeros((B, n), device=dev, dtype=torch.float32)
with sship.Homotopy(A) as h:
    del A
    torch.cuda.empty_cache()
    for name, gmin in (("gram (incl. building G)", 512), ("gram (G kept)", 512), ("gemm", 0)):
        h.set_option("batch_gram_min", gmin)
        if gmin == 0:
            h.set_option("gram_full_gib", 0)
        h.reset_stats()
        torch.cuda.synchronize()
        t0 = time.time()
        _, iters, errs = h.solve_batch(Y, 1e-3, 256, out=X)
        torch.cuda.synchronize()
        dt = time.time() - t0
        Xh = X[:64].cpu().numpy()
        ok = sum(int(np.array_equal(np.nonzero(np.abs(Xh[b]) > 1e-4 * np.abs(Xh[b]).max())[0], sups[b])) for b in range(64))
        st = h.stats()
        print("%-24s B=%d: %.3f s = %.0f signals/s, rounds %d, iters max %d, support exact %d/64, G builds %d" % (
            name, B, dt, B / dt, st["batch_rounds"], int(iters.max()), ok, st["gram_full_builds"]), flush=True)
