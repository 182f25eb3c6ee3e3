#!/usr/bin/env python3
"""GPU probe: resident lookahead kernel vs the launch-per-iteration form on small problems."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sship
from conftest import make_gaussian_problem
shapes = [(96, 700, 8), (512, 4096, 40)]
if len(sys.argv) > 1:
    shapes += [(1024, 9000, 120), (1500, 6000, 230)]
for (m, n, k) in shapes:
    A, y, x0, sup = make_gaussian_problem(5000 + m, m, n, k, np.float32)
    out = {}
    with sship.Homotopy(A) as h:
        h.set_option("trace", 1)
        for mode in (1, 2):
            h.set_option("la_fused", mode)
            t0 = time.time()
            xg, itg, eg = h.solve(y, 1e-3, 2 * k + 8)
            dt = time.time() - t0
            out[mode] = (xg.copy(), itg, eg, h.trace())
            print("shape", (m, n, k), "mode", mode, "iters", itg, "err %.3e" % eg, "%.1f ms" % (dt * 1e3),
                  "support ok", np.array_equal(np.nonzero(np.abs(xg) > 1e-4 * np.abs(xg).max())[0], sup), flush=True)
    a, b = out[1], out[2]
    print("   identical:", a[1] == b[1], np.array_equal(a[3]["idx"], b[3]["idx"]), np.array_equal(a[3]["gamma"], b[3]["gamma"]),
          np.array_equal(a[0], b[0]), "max |dx| %.3e" % np.abs(a[0] - b[0]).max(), flush=True)
