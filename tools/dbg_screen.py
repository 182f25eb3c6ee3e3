import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sship
from conftest import make_gaussian_problem
for (m, n, k) in [(1024, 8192, 40), (2048, 16384, 48)]:
    A, y, x0, sup = make_gaussian_problem(9100 + m + k, m, n, k, np.float32)
    with sship.Homotopy(A) as h:
        h.set_option("screen_single", 2)
        x, it, e = h.solve(y, 1e-3, 4 * k)
        c0, _ = h.gemv_t(y)
        order = np.argsort(-np.abs(c0), kind="stable")
        rank = {int(c): i for i, c in enumerate(order)}
        print(m, n, k, "iter", it, "worst rank of a support column in |c0|:", max(rank[int(s)] for s in sup), flush=True)
