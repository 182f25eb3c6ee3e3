#!/usr/bin/env python3
"""The screened form (csrc/screen.hip, option screen_single = 2) on random problems against the CPU oracle: fp32 and fp64,
signed coefficients, noise, both modes.  A certified signal must be the oracle's (iterations, support, coefficients to the
parity tolerance); a handed-back one must be the default engine's bit for bit.

    python tools/stress_screen.py [count]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402
import sship  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ONLY = set(int(v) for v in os.environ.get("STRESS_ONLY", "").split(",") if v.strip())
rng = np.random.default_rng(20261004)
cert = redone = bad = 0
worst = 0.0
why_total = {}
by_kind = {}
for case in range(count):
    f64 = case % 3 == 2
    dt = np.float64 if f64 else np.float32
    m = int(rng.choice([512, 768, 1024, 1536, 2048]))
    n = int(rng.choice([8192, 12000, 16384] if f64 else [2048, 4096, 8192, 16384]))
    k = int(rng.integers(4, max(5, m // 24)))
    signed = bool(rng.integers(0, 2))
    noise = float(rng.choice([0.0, 0.0, 1e-4, 1e-2]))
    fixes = bool(rng.integers(0, 2))
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(dt)
    sup = np.sort(rng.choice(n, k, replace=False))
    x0 = np.zeros(n)
    x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
    if signed:
        x0[sup] *= rng.choice([-1.0, 1.0], k)
    y = A.astype(np.float64) @ x0
    if noise:
        y = y + noise * rng.standard_normal(m)
    y = y.astype(dt)
    tol = 1e-9 if f64 else 1e-3
    budget = 3 * k + 8
    if ONLY and case not in ONLY:                     # (STRESS_ONLY="7,85": those cases only — the generator still walks every case)
        continue
    flags = oracle.SPARSE_NOTRANS | ((oracle.ZERO_ON_REMOVAL | oracle.TIE_GUARD) if fixes else 0)
    with sship.Homotopy(A, device=0) as h:
        if fixes:
            h.set_option("tie_guard", 1)
            h.set_option("zero_on_removal", 1)
        h.set_option("screen_single", 2)
        x, it, e = h.solve(y, tol, budget)
        st = h.stats()
        h.set_option("screen_single", 0)
        xd, itd, ed = h.solve(y, tol, budget)
    xo, ito, eo = oracle.homotopy(A, y, tol, budget, flags=flags)
    scale = max(1e-30, np.abs(xo).max())
    rel = np.abs(x.astype(np.float64) - xo).max() / scale
    reld = np.abs(xd.astype(np.float64) - xo).max() / scale
    certified = st["screen_signals"] == 1
    for k_, v_ in st.items():
        if k_.startswith("why_") and v_:
            why_total[k_] = why_total.get(k_, 0) + int(v_)
    kind = ("f64" if f64 else "f32", "signed" if signed else "positive", "noise %g" % noise)
    by_kind.setdefault(kind, [0, 0])
    by_kind[kind][0 if certified else 1] += 1
    cert += certified
    redone += st["screen_redone"]
    ok = True
    if certified:
        lim = (1e-10 if f64 else 1e-5)
        sig = lambda v: np.nonzero(np.abs(v) > 100 * lim * max(1e-30, np.abs(v).max()))[0]
        ok = it == ito and np.array_equal(sig(x), sig(xo)) and rel <= max(lim, 3 * reld)
        worst = max(worst, st["screen_headroom"])
    else:
        ok = it == itd and np.array_equal(x, xd)
    if not ok:
        bad += 1
    print("%3d %s m %4d n %5d k %3d signed %d noise %-6g fixes %d | %s headroom %.3f | iter %d / default %d / oracle %d | rel %.2e (default %.2e) %s"
          % (case, "f64" if f64 else "f32", m, n, k, signed, noise, fixes, "certified " if certified else "handed back", st["screen_headroom"], it, itd, ito,
             rel, reld, "" if ok else "  <-- BAD"), " ".join("%s" % k_[4:] for k_, v_ in st.items() if k_.startswith("why_") and v_), flush=True)
print("certified %d, handed back %d, bad %d of %d; largest headroom among the certified %.3f" % (cert, redone, bad, count, worst))
print("why not certified (a signal may count in several; fp64 signals count per tier):", dict(sorted(why_total.items())))
for kind in sorted(by_kind):
    print("   %-32s certified %3d  handed back %3d" % (" ".join(kind), by_kind[kind][0], by_kind[kind][1]))
sys.exit(1 if bad else 0)
