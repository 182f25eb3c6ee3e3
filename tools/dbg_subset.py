import sys, os
import numpy as np
sys.path.insert(0, "sparse-solvers_amd/python"); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
os.environ["SS_HIP_SUB_DEBUG"] = "1"
import sship
import oracle
def _batch_problem(seed, m, n, B, kmin, kmax, dtype):
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dtype)
    Y, sups = [], []
    for b in range(B):
        k = int(rng.integers(kmin, kmax + 1))
        x0 = np.zeros(n)
        sup = np.sort(rng.choice(n, k, replace=False))
        x0[sup] = 1 + np.abs(rng.standard_normal(k))
        Y.append((A.astype(np.float64) @ x0).astype(dtype))
        sups.append(sup)
    return A, np.stack(Y), sups
m, n, B, kmax_ = 128, 4096, 520, 12
A, Y, sups = _batch_problem(7000 + n + B, m, n, B, 2, kmax_, np.float32)
budget = 3 * kmax_ + 8
got = {}
with sship.Homotopy(A) as h:
    h.set_option("batch_min", 4); h.set_option("batch_gram_min", 4)
    for subset in (1, 0):
        h.set_option("batch_subset", subset)
        h.reset_stats()
        X, it, err = h.solve_batch(Y, 1e-3, budget)
        got[subset] = (X.copy(), it.copy(), err.copy())
        print("subset", subset, h.stats()["subset_signals"], h.stats()["subset_redone"], h.stats()["tie_reruns"], flush=True)
(Xs, its, es), (Xl, itl, el) = got[1], got[0]
for b in range(B):
    thr = 1e-6 * np.abs(Xl[b]).max()
    if its[b] != itl[b] or not np.array_equal(np.abs(Xs[b]) > thr, np.abs(Xl[b]) > thr):
        ds = np.nonzero((np.abs(Xs[b]) > thr) != (np.abs(Xl[b]) > thr))[0]
        xo_, ito_, eo_, tr_ = oracle.homotopy(A, Y[b], 1e-3, budget, trace=True)
        print("  differing columns", ds, "subset x", Xs[b][ds], "lockstep x", Xl[b][ds], "oracle x", xo_[ds], "true support?", [int(c in sups[b]) for c in ds])
        print("  oracle path idx", tr_["idx"][-4:], "added", tr_["added"][-4:], "gamma", tr_["gamma"][-4:], "lambda", tr_["c_inf"][-4:])
        print("  max|xs-xo|", np.abs(Xs[b]-xo_).max(), "max|xl-xo|", np.abs(Xl[b]-xo_).max(), "scale", np.abs(xo_).max())
        xo, ito, eo = oracle.homotopy(A, Y[b], 1e-3, budget)
        print("signal", b, "k", len(sups[b]), "subset it", its[b], "lockstep it", itl[b], "oracle it", ito, "nnz", (Xs[b] != 0).sum(), (Xl[b] != 0).sum(), (xo != 0).sum(),
              "err", es[b], el[b], eo)
