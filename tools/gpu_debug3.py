import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("sparse-solvers_amd/python", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import oracle, sship
from test_gpu_parity import _batch_problem
np.set_printoptions(linewidth=200, precision=7)
B = 9
A, Y, sups = _batch_problem(500 + B, 96, 640, B, 3, 9, np.float32)
h = sship.Homotopy(A)
X, iters, errs = h.solve_batch(Y, 1e-3, 40)
for b in range(B):
    xo, ito, eo = oracle.homotopy(A, Y[b], 1e-3, 40)
    x1, it1, e1 = h.solve(Y[b], 1e-3, 40)
    sup = np.nonzero(xo)[0]
    print("b", b, "k", len(sups[b]), "iters batch/single/oracle", iters[b], it1, ito, "err", errs[b], e1, eo,
          "maxdiff batch-oracle %.2e single-oracle %.2e" % (np.abs(X[b] - xo).max(), np.abs(x1 - xo).max()))
    if np.abs(X[b] - xo).max() > 1e-5:
        print("   sup", sup, "\n   xb", X[b][sup], "\n   xo", xo[sup], "\n   x1", x1[sup])
