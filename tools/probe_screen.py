#!/usr/bin/env python3
"""The screened form of one signal (csrc/screen.hip) against the default engine and the oracle: small shapes with the form
forced (option screen_single = 2), then configs[1] timed with and without it.

    python tools/probe_screen.py [--no-big] [--no-small]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402
import sship  # noqa: E402


def problem(m, n, k, seed, noise=0.0, signed=False):
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
    sup = np.sort(rng.choice(n, k, replace=False))
    x0 = np.zeros(n)
    x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
    if signed:
        x0[sup] *= rng.choice([-1.0, 1.0], k)
    y = A.astype(np.float64) @ x0
    if noise:
        y += noise * rng.standard_normal(m)
    return A, y.astype(np.float32), sup


def small():
    bad = 0
    for (m, n, k, noise, signed) in [(1024, 8192, 16, 0.0, False), (1024, 8192, 40, 0.0, False), (512, 4096, 12, 0.0, True),
                                     (1024, 8192, 24, 1e-3, True), (768, 2048, 20, 0.0, False), (2048, 16384, 48, 0.0, False)]:
        A, y, sup = problem(m, n, k, 1000 + k, noise, signed)
        xo, ito, eo = oracle.homotopy(A, y, 1e-3, 4 * k)
        for first16 in (1, 0):
            with sship.Homotopy(A, device=0) as h:
                h.set_option("screen_single", 2)
                h.set_option("screen_first16", first16)
                x, it, err = h.solve(y, 1e-3, 4 * k)
                st = h.stats()
                h.set_option("screen_single", 0)
                xd, itd, errd = h.solve(y, 1e-3, 4 * k)
            same_sup = np.array_equal(np.nonzero(x)[0], np.nonzero(xo)[0])
            rel = np.abs(x - xo).max() / max(1e-30, np.abs(xo).max())
            reld = np.abs(xd - xo).max() / max(1e-30, np.abs(xo).max())
            print("m %5d n %6d k %3d noise %g signed %d first16 %d | screened %d redone %d headroom %.3f | iter %d / default %d / oracle %d | "
                  "support %s | rel err %.2e (default engine %.2e) | err %.3e %.3e %.3e" % (
                      m, n, k, noise, signed, first16, st["screen_signals"], st["screen_redone"], st["screen_headroom"], it, itd, ito, same_sup,
                      rel, reld, err, errd, eo), flush=True)
            if it != ito or not same_sup or rel > 1e-5 + 2 * reld:
                bad += 1
    return bad


def big():
    import torch
    M, N, K = 8192, 65536, 64
    A_host = np.random.default_rng(1234).standard_normal((M, N), dtype=np.float32)
    A_host /= np.float32(np.sqrt(M))
    dev = torch.device("cuda", 0)
    A = torch.from_numpy(A_host).to(dev)
    sigs = []
    for s in range(24):
        rng = np.random.default_rng(1235 + s)
        sup = np.sort(rng.choice(N, K, replace=False))
        coef = 1.0 + np.abs(rng.standard_normal(K))
        y = (A[:, torch.from_numpy(sup).to(dev)].double() @ torch.from_numpy(coef).to(dev)).float().contiguous()
        sigs.append((y, sup, coef))
    h = sship.Homotopy(A, device=0)
    X = torch.zeros((len(sigs), N), device=dev, dtype=torch.float32)
    res = {}
    modes = ((1, 1), (1, 0), (0, 0), (1, 1))
    if os.environ.get("PROBE_MODES"):                       # e.g. "1,1;1,0": (screen_single, screen_first16) per timed round
        modes = tuple(tuple(int(v) for v in p.split(",")) for p in os.environ["PROBE_MODES"].split(";"))
    for mode, first16 in modes:
        h.set_option("screen_single", mode)
        h.set_option("screen_first16", first16)
        h.reset_stats()
        for s in range(3):
            h.solve(sigs[s][0], 1e-3, 256, out=X[s])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its = []
        for s in range(3, len(sigs)):
            _, it, err = h.solve(sigs[s][0], 1e-3, 256, out=X[s])
            its.append(it)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (len(sigs) - 3)
        st = h.stats()
        Xh = X.cpu().numpy()
        ok = sum(np.array_equal(np.nonzero(Xh[s])[0], sigs[s][1]) for s in range(len(sigs)))
        cerr = max(np.abs(Xh[s][sigs[s][1]] - sigs[s][2]).max() / sigs[s][2].max() for s in range(len(sigs)))
        res[mode] = Xh.copy()
        print("configs[1] screen_single %d first16 %d: %.4f ms per solve (%.0f signals/s), iterations %s, supports exact %d / %d, max rel coef err %.2e, "
              "screened %d redone %d headroom %.3f" % (mode, first16, dt * 1e3, 1.0 / dt, sorted(set(its)), ok, len(sigs), cerr, st["screen_signals"],
                                                      st["screen_redone"], st["screen_headroom"]), flush=True)
    if 0 in res and 1 in res:
        d = np.abs(res[1] - res[0]).max() / np.abs(res[0]).max()
        print("screened vs default engine: max |x - x'| / max |x| = %.2e" % d)
    # profiled solves: where the time goes
    h.set_option("screen_single", 1)
    h.set_profiling(True)
    h.reset_stats()
    for s in range(3, 13):
        h.solve(sigs[s][0], 1e-3, 256, out=X[s])
    st = h.stats()
    h.set_profiling(False)
    if st["screen_launches"]:
        ms = st["screen_ms"] / st["screen_launches"]
        print("screening pass: %.4f ms, %.0f GB/s of %d bytes; A^T y %.4f ms; solve %.4f ms" % (
            ms, st["screen_bytes"] / st["screen_launches"] / (ms * 1e-3) / 1e9, st["screen_bytes"] // st["screen_launches"],
            st["sweep1_ms"] / max(1, st["sweep1_launches"]), st["solve_ms"] / max(1, st["solves"])))
    h.close()


if __name__ == "__main__":
    bad = 0
    if "--no-small" not in sys.argv:
        bad = small()
        print("small shapes: %d bad" % bad, flush=True)
    if "--no-big" not in sys.argv:
        big()
    sys.exit(1 if bad else 0)
