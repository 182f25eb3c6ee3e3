import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
m, n, k, seed = 2048, 16384, 60, 2
rng = np.random.default_rng(5000 + seed)
A = rng.standard_normal((m, n)) / np.sqrt(m)
sup = np.sort(rng.choice(n, k, replace=False))
x0 = np.zeros(n); x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
y = A @ x0
with sship.Homotopy(A, device=0) as h:
    h.set_option("screen_single", 2)
    x, it, err = h.solve(y, 1e-9, 4 * k)
    print(it, err, h.stats()["screen_signals"], h.stats()["screen_redone"])
