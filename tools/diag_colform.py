"""Column form vs the other batch forms vs the oracle on removal-heavy problems (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle, sship
from test_gpu_parity import _batch_problem
A, Y, sups = _batch_problem(77, 96, 400, 48, 6, 14, np.float32)
res = {}
with sship.Homotopy(A) as h:
    for label, opts in (("cols", {}), ("seq", {"batch_cols_min": 0}), ("gemm", {"batch_cols_min": 0, "batch_min": 4}),
                        ("gram", {"batch_cols_min": 0, "batch_min": 4, "batch_gram_min": 4})):
        for k_, v in opts.items():
            h.set_option(k_, v)
        X, iters, errs = h.solve_batch(Y, 1e-3, 60)
        res[label] = (X.copy(), iters.copy())
        print(label, h.stats()["batch_col_rounds"], h.stats()["batch_rounds"])
        h.reset_stats()
orc = [oracle.homotopy(A, y, 1e-3, 60, trace=True) for y in Y]
for b in range(48):
    xo, ito, eo, tro = orc[b]
    rem = int((tro["added"][:ito + 1] == 0).sum())
    line = "b %2d oracle it %2d rem %d |" % (b, ito, rem)
    for label in ("cols", "seq", "gemm", "gram"):
        X, iters = res[label]
        ok = int(iters[b]) == ito and np.array_equal(X[b] != 0, xo != 0)
        line += " %s it %2d %s" % (label, iters[b], "ok " if ok else "DIFF")
    print(line)
