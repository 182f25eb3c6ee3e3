#!/usr/bin/env python3
"""GPU probe: two solver contexts used from two host threads at once (resident grids compete for the CUs)."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sship
from conftest import make_gaussian_problem
m, n, k = 2048, 65536, 32
probs = [make_gaussian_problem(900 + t, m, n, k, np.float32) for t in range(2)]
hs = [sship.Homotopy(p[0]) for p in probs]
results = [None, None]

def work(t):
    A, y, x0, sup = probs[t]
    ok, tmax = 0, 0.0
    for r in range(12):
        t0 = time.time()
        x, it, e = hs[t].solve(y, 1e-3, 4 * k)
        tmax = max(tmax, time.time() - t0)
        ok += int(np.array_equal(np.nonzero(np.abs(x) > 1e-4 * np.abs(x).max())[0], sup))
    results[t] = (ok, tmax, hs[t].stats()["persist_fallbacks"], hs[t].get_option("la_fused"))

for t in range(2):                       # warm-up, one at a time
    hs[t].solve(probs[t][1], 1e-3, 4 * k)
t0 = time.time()
ths = [threading.Thread(target=work, args=(t,)) for t in range(2)]
[t.start() for t in ths]
[t.join() for t in ths]
print("both threads done in %.2f s" % (time.time() - t0))
for t in range(2):
    print("thread", t, "solves exact %d/12, slowest solve %.3f s, fallbacks %d, la_fused now %d" % results[t])
