#!/usr/bin/env python3
"""GPU probe: the fp64 lookahead sweep (k_gemm32_tn_f64) on constant vs random data at the configs[4] shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, torch, sship
m, n = 16384, 131072
b = m * n * 8 + 32 * m * 8 + 32 * n * 8
cols = np.arange(0, 32000, 1000, dtype=np.uint32)
for kind in ("constant", "random", "constant", "random"):
    if kind == "constant":
        A = torch.full((m, n), 0.0115, device="cuda", dtype=torch.float64)
    else:
        g = torch.Generator(device="cuda").manual_seed(3)
        A = torch.randn((m, n), generator=g, device="cuda", dtype=torch.float64); A /= np.sqrt(m)
    with sship.Homotopy(A) as h:
        del A; torch.cuda.empty_cache()
        for reps in (1, 5):
            _, ms = h.gram_cols(cols, reps)
            print("%-8s data, %d launch(es) back to back: %.3f ms per launch = %.0f GB/s" % (kind, reps, ms, b / ms / 1e6), flush=True)
