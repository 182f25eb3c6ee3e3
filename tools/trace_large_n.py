#!/usr/bin/env python3
"""GPU probe for rocprofv3: two default-form solves at n = 1M columns (kernel timeline of the second)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, torch, sship
m, n, k = 4096, 1 << 20, 48
g = torch.Generator(device="cuda").manual_seed(99)
A = torch.randn((m, n), generator=g, device="cuda", dtype=torch.float32) / np.sqrt(m)
rng = np.random.default_rng(100)
sup = np.sort(rng.choice(n, k, replace=False)); coef = 1.0 + np.abs(rng.standard_normal(k))
y = (A[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).cuda()).float()
with sship.Homotopy(A) as h:
    del A; torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        h.set_option("la_fused", int(sys.argv[1]))
    for _ in range(2):
        x, it, e = h.solve(y, 1e-3, 4 * k)
    print("iters", it, h.stats()["solo_solves"], h.stats()["solo_retries"])
