#!/usr/bin/env python3
"""GPU probe: OMP at the C2 size, residual form vs Gram form."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship, torch
m, n, k = 8192, 65536, 64
g = torch.Generator(device="cuda:0").manual_seed(1234)
A = torch.randn((m, n), generator=g, device="cuda:0", dtype=torch.float32) / np.sqrt(m)
rng = np.random.default_rng(1)
sup = np.sort(rng.choice(n, k, replace=False))
coef = (1.0 + np.abs(rng.standard_normal(k))).astype(np.float32)
y = (A[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).double().cuda()).float().contiguous()
with sship.Homotopy(A) as h:
    for eng in (1, 0, 1):
        h.set_option("engine", eng)
        h.reset_stats()
        torch.cuda.synchronize()
        t0 = time.time()
        x, it, e = h.solve_omp(y, 1e-3, 256)
        dt = time.time() - t0
        ok = np.array_equal(np.nonzero(x)[0], sup)
        print("engine %d: %.2f ms, iters %d, support exact %s, max coef err %.2e, lookahead sweeps %d" % (
            eng, dt * 1e3, it, ok, np.abs(x[sup] - coef).max() / coef.max(), h.stats()["lookahead_sweeps"]), flush=True)
