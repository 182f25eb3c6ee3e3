#!/usr/bin/env python3
"""GPU probe: coefficient error of the three fp32 engines against the fp32 and fp64 oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sship, oracle
from conftest import make_gaussian_problem
ENG = {"sweep": {"engine": 0}, "la": {"engine": 1, "la_fused": 0}, "la-fused": {"engine": 1, "la_fused": 1}}
for (m, n, k) in [(96, 700, 8), (256, 3000, 20), (1024, 9000, 48), (300, 1500, 20)]:
    A, y, x0, sup = make_gaussian_problem(4000 + m, m, n, k, np.float32)
    xo, ito, eo = oracle.homotopy(A, y, 1e-3, 4 * k)[:3]
    xd, itd, ed = oracle.homotopy(A.astype(np.float64), y.astype(np.float64), 1e-3, 4 * k)[:3]
    sc = np.abs(xd).max()
    print("shape", (m, n, k), "iters f32/f64", ito, itd, "oracle32 vs 64: %.2e" % (np.abs(xo - xd).max() / sc))
    with sship.Homotopy(A) as h:
        for name, opts in ENG.items():
            for kk, v in opts.items():
                h.set_option(kk, v)
            xg, itg, eg = h.solve(y, 1e-3, 4 * k)
            print("   %-9s it %d  vs o32 %.2e  vs o64 %.2e" % (name, itg, np.abs(xg - xo).max() / sc, np.abs(xg - xd).max() / sc))
