#!/usr/bin/env python3
"""GPU probe: C2 single-signal solves with G = A^T A as the cache (option gram_full_after)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship, torch
m, n, k = 8192, 65536, 64
g = torch.Generator(device="cuda:0").manual_seed(1234)
A = torch.randn((m, n), generator=g, device="cuda:0", dtype=torch.float32) / np.sqrt(m)
rng = np.random.default_rng(3)
def signal():
    sup = np.sort(rng.choice(n, k, replace=False))
    coef = (1.0 + np.abs(rng.standard_normal(k))).astype(np.float32)
    y = (A[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).double().cuda()).float().contiguous()
    return y, sup, coef
sigs = [signal() for _ in range(12)]
with sship.Homotopy(A) as h:
    for label, after in (("cache + lookahead sweeps", 0), ("G = A^T A in HBM", 1)):
        h.set_option("gram_full_after", after)
        ts = []
        ok = 0
        for (y, sup, coef) in sigs:
            torch.cuda.synchronize()
            t0 = time.time()
            x, it, e = h.solve(y, 1e-3, 256)
            ts.append(time.time() - t0)
            ok += int(np.array_equal(np.nonzero(x)[0], sup) and np.abs(x[sup] - coef).max() <= 1e-5 * coef.max())
        st = h.stats()
        print("%-26s first %.1f ms, then median %.3f ms (%.0f signals/s); exact %d/12; G builds %d, lookahead sweeps %d" % (
            label, ts[0] * 1e3, np.median(ts[1:]) * 1e3, 1.0 / np.median(ts[1:]), ok, st["gram_full_builds"], st["lookahead_sweeps"]), flush=True)
    t0 = time.time(); x, it, e = h.solve_omp(sigs[0][0], 1e-3, 256); print("omp with G: %.3f ms, iters %d" % ((time.time() - t0) * 1e3, it))
