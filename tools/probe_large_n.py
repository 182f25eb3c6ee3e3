#!/usr/bin/env python3
"""GPU probe: a wide dictionary (n = 1M columns, m = 4096, fp32, 16 GiB) through every single-signal form."""
import os, sys, time
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
    ref = None
    for name, opts in (("resident/iter", {"engine": 1, "la_fused": 2}), ("speculative", {"la_fused": 3}), ("launch per iteration", {"la_fused": 1}), ("sweep per iteration", {"engine": 0})):
        for kk, v in opts.items():
            h.set_option(kk, v)
        h.reset_stats()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, it, e = h.solve(y, 1e-3, 4 * k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = h.stats()
        ok = np.array_equal(np.nonzero(x)[0], sup)
        print("%-22s %8.2f ms  iters %d  exact %s  lookahead sweeps %d  solo %d/%d" % (name, dt * 1e3, it, ok, st["lookahead_sweeps"], st["solo_solves"], st["solo_retries"]), flush=True)
        if ref is None: ref = x
        else: print("      max |x - x_first| = %.2e" % np.abs(x - ref).max())
