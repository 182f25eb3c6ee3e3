#!/usr/bin/env python3
"""GPU probe for rocprofv3: C2-size solves in the speculative form (la_fused 3); kernel timeline of the last."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, torch, sship
m, n, k = 8192, 65536, 64
g = torch.Generator(device="cuda").manual_seed(1234)
A = torch.randn((m, n), generator=g, device="cuda", dtype=torch.float32) / np.sqrt(m)
rng = np.random.default_rng(1235)
sup = np.sort(rng.choice(n, k, replace=False)); coef = 1.0 + np.abs(rng.standard_normal(k))
y = (A[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).cuda()).float()
with sship.Homotopy(A) as h:
    h.set_option("la_fused", int(sys.argv[1]) if len(sys.argv) > 1 else 3)
    for _ in range(4):
        x, it, e = h.solve(y, 1e-3, 256)
    print("iters", it, h.stats()["solo_solves"], h.stats()["solo_retries"])
