#!/usr/bin/env python3
"""GPU stress probe: many random problems (shape, sparsity, noise, tolerance, subset size), speculative form
against the resident form: iterations, path and coefficients must agree bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, sship
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
bad = fails = solo = 0
t0 = time.time()
for case in range(ncase):
    m = int(rng.choice([16, 40, 64, 128, 300, 700, 1500]))
    n = int(rng.choice([64, 700, 3000, 20000, 20000, 33000, 70000, 70000]))
    k = int(rng.integers(1, max(2, min(m // 2, 90))))
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
    x0 = np.zeros(n, np.float32)
    x0[rng.choice(n, min(k, n), replace=False)] = (1 + np.abs(rng.standard_normal(min(k, n)))) * rng.choice([-1.0, 1.0], min(k, n))
    y = (A @ x0 + float(rng.choice([0.0, 0.0, 0.01, 0.05])) * rng.standard_normal(m)).astype(np.float32)
    tol = float(rng.choice([1e-4, 1e-3, 1e-2, 5e-2]))
    max_iter = int(min(3 * m, 250))
    with sship.Homotopy(A) as h:
        h.set_option("engine", 2)
        h.set_option("trace", 1)
        h.set_option("la_fused", 2)
        x2, it2, e2 = h.solve(y, tol, max_iter); t2 = h.trace()
        h.set_option("la_fused", 3)
        h.set_option("solo_subset", int(rng.choice([256, 256, 256, 60, 10])))
        # (n > 16384: the early form; its switches decide which Gram columns are fetched when, never a result)
        h.set_option("early_solo", int(rng.choice([1, 1, 1, 0])))
        h.set_option("early_pass", int(rng.choice([2, 2, 0])))
        h.set_option("early_adapt", int(rng.choice([1, 1, 0])))
        h.reset_stats()
        x3, it3, e3 = h.solve(y, tol, max_iter); t3 = h.trace()
        x3b, it3b, e3b = h.solve(y, tol, max_iter)           # and again on the same context
        st = h.stats()
    solo += st["solo_solves"]; fails += st["solo_retries"]
    same = (it2 == it3 == it3b and np.array_equal(t2["idx"], t3["idx"]) and np.array_equal(t2["gamma"], t3["gamma"])
            and np.array_equal(x2, x3, equal_nan=True) and np.array_equal(x2, x3b, equal_nan=True))
    if not same:
        bad += 1
        print("MISMATCH case %d: m %d n %d k %d tol %g iters %d/%d/%d" % (case, m, n, k, tol, it2, it3, it3b), flush=True)
print("%d cases in %.1f s: %d mismatches, %d speculative solves, %d failed checks" % (ncase, time.time() - t0, bad, solo, fails))
sys.exit(1 if bad else 0)
