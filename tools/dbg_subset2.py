import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
os.environ["SS_HIP_SUB_DEBUG"] = "1"
import sship
rng = np.random.default_rng(5)
m, n, B, kmax = 400, 9000, 300, 50
A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
Y = []
ks = []
for b in range(B):
    k = int(rng.integers(2, kmax + 1)); ks.append(k)
    x0 = np.zeros(n); x0[rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
    Y.append((A.astype(np.float64) @ x0).astype(np.float32))
Y = np.stack(Y)
with sship.Homotopy(A) as h:
    if os.environ.get("FIXES", "1") == "1":
        h.set_option("tie_guard", 1); h.set_option("zero_on_removal", 1)
    h.set_option("batch_min", 4); h.set_option("batch_gram_min", 4)
    X, it, err = h.solve_batch(Y, 1e-3, 70)
    print("accepted", h.stats()["subset_signals"], "redone", h.stats()["subset_redone"], "k of first 20", ks[:20], "iters", it[:20])
