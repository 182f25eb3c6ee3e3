"""Developer aid: why does a reference-mode (tie_guard 0, zero_on_removal 0) solve diverge from the oracle?
Prints, for one of the ill-conditioned 40 x 120 fp32 problems of test_engines_agree_on_removal_paths, the
first breakpoint at which the device path and the oracle's path differ and what happens after it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle
import sship

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2013
rng = np.random.default_rng(seed)
m, n, k = 40, 120, 14
A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
x0 = np.zeros(n, np.float32)
x0[rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
y = A @ x0
xo, ito, eo, tro = oracle.homotopy(A, y, 1e-3, 200, trace=True)
print("oracle: iter", ito, "err", eo)
with sship.Homotopy(A) as h:
    h.set_option("trace", 1)
    for eng, fused in ((0, 0), (1, 1), (1, 2), (1, 3)):
        h.set_option("engine", eng)
        h.set_option("la_fused", fused)
        xg, itg, eg = h.solve(y, 1e-3, 200)
        tg = h.trace()
        nb = min(len(tg["idx"]), len(tro["idx"]))
        diff = next((t for t in range(nb) if tg["idx"][t] != tro["idx"][t] or tg["added"][t] != tro["added"][t]), None)
        print("engine", eng, "la_fused", fused, "iter", itg, "err", eg, "first differing breakpoint", diff)
        if diff is not None:
            lo, hi = max(0, diff - 2), min(len(tg["idx"]), diff + 8)
            print("  device:", [(int(i), int(a), float("%.3g" % g)) for i, a, g in zip(tg["idx"][lo:hi], tg["added"][lo:hi], tg["gamma"][lo:hi])])
            hi2 = min(len(tro["idx"]), diff + 8)
            print("  oracle:", [(int(i), int(a), float("%.3g" % g)) for i, a, g in zip(tro["idx"][lo:hi2], tro["added"][lo:hi2], tro["gamma"][lo:hi2])])
            print("  device tail:", [(int(i), int(a), float("%.3g" % g)) for i, a, g in zip(tg["idx"][-6:], tg["added"][-6:], tg["gamma"][-6:])])
