#!/usr/bin/env python3
"""GPU probe for rocprofv3: one lock-step batch in Gram form after G exists."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship, torch
m, n, k = 8192, 65536, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1234)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32) / np.sqrt(m)
rng = np.random.default_rng(7)
Y = torch.empty((B, m), device=dev, dtype=torch.float32)
for b in range(B):
    sup = np.sort(rng.choice(n, k, replace=False))
    coef = torch.from_numpy((1.0 + np.abs(rng.standard_normal(k))).astype(np.float32)).to(dev)
    Y[b] = A[:, torch.from_numpy(sup).to(dev)] @ coef
X = torch.zeros((B, n), device=dev, dtype=torch.float32)
with sship.Homotopy(A) as h:
    del A
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        h.solve_batch(Y, 1e-3, 256, out=X)
        torch.cuda.synchronize(); print("batch %d: %.3f s" % (rep, time.time() - t0), flush=True)
