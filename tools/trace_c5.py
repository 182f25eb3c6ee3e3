#!/usr/bin/env python3
"""GPU probe for rocprofv3: two fp64 solves at the configs[4] shape (kernel timeline of the second)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship, torch
m, n, k = 16384, 131072, 128
g = torch.Generator(device="cuda:0").manual_seed(4321)
A = torch.randn((m, n), generator=g, device="cuda:0", dtype=torch.float64)
A /= np.sqrt(m)
rng = np.random.default_rng(4322)
sup = np.sort(rng.choice(n, k, replace=False))
coef = 1.0 + np.abs(rng.standard_normal(k))
y = (A[:, torch.from_numpy(sup).to("cuda:0")] @ torch.from_numpy(coef).to("cuda:0")).contiguous()
with sship.Homotopy(A) as h:
    del A
    torch.cuda.empty_cache()
    for _ in range(2):
        x, it, err = h.solve(y, 1e-9, 512)
    print("iters", it)
