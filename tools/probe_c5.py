#!/usr/bin/env python3
"""GPU probe: BASELINE configs[4] (A 16384 x 131072 fp64, k = 128): lookahead engine vs one sweep per iteration."""
import os, sys, time
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
    res = {}
    for name, opts in (("lookahead", {"engine": 1}), ("sweep/iteration", {"engine": 0}), ("lookahead", {"engine": 1})):
        for kk, v in opts.items():
            h.set_option(kk, v)
        h.reset_stats()
        torch.cuda.synchronize()
        t0 = time.time()
        x, it, err = h.solve(y, 1e-9, 512)
        dt = time.time() - t0
        ok = np.array_equal(np.nonzero(x)[0], sup) and np.abs(x[sup] - coef).max() <= 1e-10 * coef.max()
        st = h.stats()
        print("%-16s %8.2f ms  iters %d  exact %s  lookahead sweeps %d" % (name, dt * 1e3, it, ok, st["lookahead_sweeps"]), flush=True)
        res[name] = x
    print("max |x_la - x_sweep| = %.2e" % np.abs(res["lookahead"] - res["sweep/iteration"]).max())
    cols = np.arange(0, 32000, 1000, dtype=np.uint32)
    G, ms = h.gram_cols(cols, 5)
    b = m * n * 8 + 32 * m * 8 + 32 * n * 8
    print("fp64 lookahead sweep: %.3f ms = %.0f GB/s (%.0f%% of 8 TB/s), %.1f TFLOP/s" % (ms, b / ms / 1e6, b / ms / 1e6 / 80, 2.0 * m * n * 32 / ms / 1e9))
    r = rng.standard_normal(m)
    c, ms1 = h.gemv_t(r, 3)
    print("fp64 1-RHS sweep: %.3f ms = %.0f GB/s" % (ms1, (m * n * 8 + m * 8 + n * 8) / ms1 / 1e6))
