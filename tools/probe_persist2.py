#!/usr/bin/env python3
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sship
from conftest import make_gaussian_problem
A, y, _, _ = make_gaussian_problem(9, 64, 256, 8, np.float32)
for fresh in (True, False):
    h = sship.Homotopy(A)
    for mi in (1, 2, 5, 40):
        for mode in (1, 2):
            h.set_option("la_fused", mode)
            h.set_option("trace", 1)
            t0 = time.time()
            try:
                xg, itg, eg = h.solve(y, 1e-3, mi)
                print("mi", mi, "mode", mode, "ok iters", itg, "%.1f ms" % ((time.time() - t0) * 1e3), h.trace()["idx"][:8], flush=True)
            except Exception as e:
                print("mi", mi, "mode", mode, "FAILED", e, "%.1f ms" % ((time.time() - t0) * 1e3), flush=True)
        if fresh:
            h.close() if hasattr(h, "close") else None
            h = sship.Homotopy(A)
