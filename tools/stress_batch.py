#!/usr/bin/env python3
"""GPU stress probe: mid-size batches in the column form on random problems (shape, batch size, sparsity, noise,
tolerance, both modes) against the CPU oracle signal by signal, and against the one-solve-per-signal path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, sship, oracle
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
tot = agree = agree_seq = bad = 0
worst = 0.0
t0 = time.time()
for case in range(ncase):
    m = int(rng.choice([48, 96, 200, 400]))
    n = int(rng.choice([300, 1000, 2500, 6000]))
    B = int(rng.integers(24, 140))
    kmax = max(3, m // 6)
    noise = float(rng.choice([0.0, 0.0, 0.01]))
    tol = float(rng.choice([1e-3, 1e-2]))
    fixes = int(rng.integers(0, 2))
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
    Y = []
    for b in range(B):
        k = int(rng.integers(2, kmax + 1))
        x0 = np.zeros(n)
        x0[rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
        Y.append((A.astype(np.float64) @ x0 + noise * rng.standard_normal(m)).astype(np.float32))
    Y = np.stack(Y)
    max_iter = int(min(2 * m, 120))
    flags = oracle.SPARSE_NOTRANS | ((oracle.ZERO_ON_REMOVAL | oracle.TIE_GUARD) if fixes else 0)
    with sship.Homotopy(A) as h:
        if fixes:
            h.set_option("tie_guard", 1); h.set_option("zero_on_removal", 1)
        X, iters, errs = h.solve_batch(Y, tol, max_iter)
        assert h.stats()["batch_col_rounds"] > 0
        h.set_option("batch_cols_min", 0)
        Xs, iters_s, errs_s = h.solve_batch(Y, tol, max_iter)
    A64 = A.astype(np.float64)
    for b in range(0, B, 3):
        xo, ito, eo = oracle.homotopy(A, Y[b], tol, max_iter, flags=flags)
        tot += 1
        if int(iters_s[b]) == ito:
            agree_seq += 1
        if int(iters[b]) == ito:
            agree += 1
            if ito >= max_iter:
                continue                         # (a path that ran to the budget: rounding-chaotic in fp32 on the CPU too)
            # The yardstick is the reference algorithm itself in fp32 against fp64 on the same data: a device path must be
            # as close to the double-precision answer as the fp32 oracle is.  (Distance to the fp32 ORACLE is not one: the
            # last step of a noise-free path — every column ties at lambda -> 0 — lands where the largest rounding error
            # among n candidates puts it, 0.5-1 % short of the fp64 step on every fp32 implementation, the oracle
            # included; two of them agreeing there to 1e-5 is luck.  tools/dbg_colform.py)
            xd, itd, ed = oracle.homotopy(A64, Y[b].astype(np.float64), tol, max_iter, flags=flags)
            scale = max(1.0, np.abs(xd).max())
            er = np.abs(xo - xd).max() / scale
            dc = np.abs(X[b] - xd).max() / scale
            ds = np.abs(Xs[b] - xd).max() / scale if int(iters_s[b]) == ito else 0.0
            if ds > 0.0:
                worst = max(worst, dc / max(ds, 1e-7))
            # (floor 2e-4: Gram-form correlations carry an absolute error ~eps * ||c0||_inf * sqrt(K); on m <= 100 problems that
            # is ~1e-4 on the coefficients' scale — for the one-solve path, also Gram form, alike: the ratio of the two is printed)
            if dc > max(5.0 * er, 2e-4):
                bad += 1
                print("COEFFICIENTS case %d signal %d: m %d n %d B %d tol %g iters %d  |x - x_fp64|: fp32 oracle %.3g  column form %.3g  one solve %.3g" % (
                    case, b, m, n, B, tol, ito, er, dc, ds), flush=True)
print("%d cases, %d signals checked in %.1f s: column form agrees with the oracle's iteration count on %d (one solve per signal: %d), "
      "%d signals further from the fp64 answer than max(5 x the fp32 oracle's distance, 2e-4) (worst column-form / one-solve distance ratio %.2f)" % (ncase, tot, time.time() - t0, agree, agree_seq, bad, worst))
sys.exit(1 if bad or agree < 0.9 * tot else 0)
