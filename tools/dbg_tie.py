"""developer probe: which forms raise the tie-stall flag on a given problem"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "sparse-solvers_amd", "python")]
import sship, oracle
from conftest import make_gaussian_problem
m, n, k = 1024, 9000, 120
A, y, x0, sup = make_gaussian_problem(5000 + m, m, n, k, np.float32)
xo, ito, eo, tro = oracle.homotopy(A, y, 1e-3, 2 * k + 8, trace=True)
print("oracle iter", ito, "err", eo, "min gamma", tro["gamma"][1:].min())
with sship.Homotopy(A) as h:
    h.set_option("trace", 1)
    for name, opts in (("sweep", {"engine": 0}), ("la0", {"engine": 1, "la_fused": 0}), ("la1", {"engine": 1, "la_fused": 1}),
                       ("la2", {"engine": 1, "la_fused": 2}), ("la3", {"engine": 1, "la_fused": 3}), ("ro", {"engine": 3})):
        for kk, v in opts.items():
            h.set_option(kk, v)
        for rerun in (1, 0):
            h.set_option("tie_rerun", rerun)
            h.reset_stats()
            xg, itg, eg = h.solve(y, 1e-3, 2 * k + 8)
            st = h.stats()
            tr = h.trace()
            same = np.array_equal(tr["idx"][:-1], tro["idx"][:-1])
            print(name, "rerun", rerun, "iter", itg, "tie_reruns", st["tie_reruns"], "solo", st["solo_solves"], st["solo_retries"],
                  "path==oracle", same, "maxdiff", np.abs(xg - xo).max())
