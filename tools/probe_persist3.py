#!/usr/bin/env python3
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sship
from conftest import make_gaussian_problem
m, n, k = 1024, 9000, 120
A, y, x0, sup = make_gaussian_problem(5000 + m, m, n, k, np.float32)
with sship.Homotopy(A) as h:
    for mode in (1, 2, 2, 2):
        h.set_option("la_fused", mode)
        h.reset_stats() if hasattr(h, "reset_stats") else None
        t0 = time.time()
        xg, itg, eg = h.solve(y, 1e-3, 2 * k + 8)
        print("mode", mode, "iters", itg, "%.2f ms" % ((time.time() - t0) * 1e3), h.stats(), flush=True)
