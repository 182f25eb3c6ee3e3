#!/usr/bin/env python3
"""GPU-box diagnostic: repeat solves of a few seeded problems, compare the device's homotopy
path with the oracle's and with itself across repetitions (a run-to-run difference means a
race)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("sparse-solvers_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import oracle  # noqa: E402
import sship  # noqa: E402
from conftest import make_gaussian_problem  # noqa: E402

cases = [(7, 96, 700, 9, np.float64, 1e-9, 60), (7, 96, 700, 9, np.float32, 1e-3, 60),
         (10, 128, 2048, 12, np.float32, 1e-3, 64), (612, 512, 4096, 24, np.float64, 1e-9, 96)]
bad = 0
for seed, m, n, k, dt, tol, mi in cases:
    A, y, x0, sup = make_gaussian_problem(seed, m, n, k, dt)
    xo, ito, eo, tro = oracle.homotopy(A, y, tol, mi, trace=True)
    for view in (A, np.asfortranarray(A)):
        h = sship.Homotopy(view)
        h.set_option("trace", 1)
        first = None
        for rep in range(40):
            x, it, err = h.solve(y, tol, mi)
            tr = h.trace()
            key = (it, x.tobytes(), tr["idx"].tobytes())
            if first is None:
                first = key
            same_self = key == first
            same_path = it == ito and np.array_equal(tr["idx"][:-1], tro["idx"][:-1]) and \
                np.array_equal(tr["added"][:len(tro["added"])], tro["added"])
            same_sup = np.array_equal(np.nonzero(x)[0], np.nonzero(xo)[0])
            if not (same_self and same_path and same_sup):
                bad += 1
                print("MISMATCH case", (seed, m, n, k, dt.__name__), "rep", rep, "self", same_self,
                      "path", same_path, "support", same_sup, "iter", it, ito)
                print("  gpu idx  ", tr["idx"].tolist())
                print("  ora idx  ", tro["idx"].tolist())
                print("  gpu added", tr["added"].tolist())
                print("  gpu gamma", np.array2string(tr["gamma"], precision=6))
                print("  ora gamma", np.array2string(tro["gamma"], precision=6))
                print("  extra nz ", sorted(set(np.nonzero(x)[0]) ^ set(np.nonzero(xo)[0])),
                      [float(x[i]) for i in sorted(set(np.nonzero(x)[0]) ^ set(np.nonzero(xo)[0]))])
                if bad > 6:
                    sys.exit(1)
        h.close()
print("done, mismatches:", bad)
