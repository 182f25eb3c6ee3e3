#!/usr/bin/env python3
"""GPU stress probe: batches in the SUBSET form of the Gram form (csrc/subbatch.hip) on random problems (shape, batch size,
sparsity, noise, tolerance, both modes) against the CPU oracle signal by signal, and against the lock-step form."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, sship, oracle
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
tot = agree = agree_l = bad = acc = red = ties = 0
t0 = time.time()
for case in range(ncase):
    m = int(rng.choice([96, 200, 400, 800]))
    n = int(rng.choice([400, 1500, 4000, 9000]))
    B = int(rng.integers(150, 500))
    kmax = max(3, m // 8)
    noise = float(rng.choice([0.0, 0.0, 0.01]))
    tol = float(rng.choice([1e-3, 1e-2]))
    fixes = int(rng.integers(0, 2))
    signed = int(rng.integers(0, 2))
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
    Y = []
    for b in range(B):
        k = int(rng.integers(2, kmax + 1))
        x0 = np.zeros(n)
        x0[rng.choice(n, k, replace=False)] = (1 + np.abs(rng.standard_normal(k))) * (rng.choice([-1.0, 1.0], k) if signed else 1.0)
        Y.append((A.astype(np.float64) @ x0 + noise * rng.standard_normal(m)).astype(np.float32))
    Y = np.stack(Y)
    max_iter = int(min(2 * m, 70))
    flags = oracle.SPARSE_NOTRANS | ((oracle.ZERO_ON_REMOVAL | oracle.TIE_GUARD) if fixes else 0)
    with sship.Homotopy(A) as h:
        if fixes:
            h.set_option("tie_guard", 1); h.set_option("zero_on_removal", 1)
        h.set_option("batch_min", 4); h.set_option("batch_gram_min", 4)
        h.reset_stats()
        X, iters, errs = h.solve_batch(Y, tol, max_iter)
        st = h.stats()
        acc += int(st["subset_signals"]); red += int(st["subset_redone"]); ties += int(st["tie_reruns"])
        h.set_option("batch_subset", 0)
        Xl, iters_l, errs_l = h.solve_batch(Y, tol, max_iter)
    A64 = A.astype(np.float64)
    for b in range(0, B, 5):
        xo, ito, eo = oracle.homotopy(A, Y[b], tol, max_iter, flags=flags)
        tot += 1
        agree_l += int(iters_l[b]) == ito
        if int(iters[b]) == ito:
            agree += 1
            if ito >= max_iter:
                continue
            xd, itd, ed = oracle.homotopy(A64, Y[b].astype(np.float64), tol, max_iter, flags=flags)
            scale = max(1.0, np.abs(xd).max())
            er = np.abs(xo - xd).max() / scale
            dc = np.abs(X[b] - xd).max() / scale
            dl = np.abs(Xl[b] - xd).max() / scale
            if dc > max(5.0 * er, 2e-4) and dc > 2.0 * dl:            # (further out than the lock-step form of the same signal, too)
                bad += 1
                print("COEFFICIENTS case %d signal %d: m %d n %d B %d tol %g iters %d  |x - x_fp64|: fp32 oracle %.3g  subset form %.3g  lock-step %.3g" % (
                    case, b, m, n, B, tol, ito, er, dc, dl), flush=True)
    print("case %d: m %d n %d B %d noise %g tol %g fixes %d signed %d: accepted %d redone %d ties %d" % (case, m, n, B, noise, tol, fixes, signed, int(st["subset_signals"]), int(st["subset_redone"]), int(st["tie_reruns"])), flush=True)
print("%d cases, %d signals checked in %.1f s: the subset form agrees with the oracle's iteration count on %d (lock-step form: %d); %d signals further "
      "from the fp64 answer than max(5 x the fp32 oracle's distance, 2e-4); %d accepted, %d redone in lock-step, %d tie re-runs" % (
          ncase, tot, time.time() - t0, agree, agree_l, bad, acc, red, ties))
sys.exit(1 if bad or agree < agree_l - max(2, tot // 100) else 0)
