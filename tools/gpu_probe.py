#!/usr/bin/env python3
"""Development probe (GPU box): times every sweep variant at the headline shape and one
full solve.  Not part of the test-suite or the bench contract."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=8192)
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7,8,9,10,11")
    ap.add_argument("--repeats", type=int, default=30)
    ap.add_argument("--solves", type=int, default=3)
    args = ap.parse_args()
    import torch
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    npdt = np.float32 if args.dtype == "f32" else np.float64
    m, n, k = args.m, args.n, args.k
    g = torch.Generator(device="cuda:0").manual_seed(1234)
    A = torch.randn((m, n), generator=g, device="cuda:0", dtype=dt) / np.sqrt(m)
    rng = np.random.default_rng(1235)
    sup = np.sort(rng.choice(n, k, replace=False))
    coef = 1.0 + np.abs(rng.standard_normal(k))
    y = (A[:, torch.from_numpy(sup).to("cuda:0")].double() @ torch.from_numpy(coef).to("cuda:0")).to(dt).contiguous()
    t0 = time.time()
    h = sship.Homotopy(A)
    torch.cuda.synchronize()
    print("create: %.3f s" % (time.time() - t0), flush=True)
    s = np.dtype(npdt).itemsize
    bytes1 = m * n * s + m * s + n * s
    r = rng.standard_normal(m).astype(npdt)
    out = {}
    for v in [int(x) for x in args.variants.split(",")]:
        h.set_option("sweep_variant", v)
        h.gemv_t(r, 3)
        best = 1e9
        for _ in range(3):
            c, ms = h.gemv_t(r, args.repeats)
            best = min(best, ms)
        out[v] = best
        print("variant %2d: sweep1 %.4f ms  %.1f GB/s" % (v, best, bytes1 / best / 1e6), flush=True)
    bestv = min(out, key=out.get)
    print("best variant", bestv, flush=True)
    h.set_option("sweep_variant", bestv)
    h.set_profiling(True)
    tol = 1e-3 if args.dtype == "f32" else 1e-9
    for i in range(args.solves):
        h.reset_stats()
        t0 = time.time()
        x, it, err = h.solve(y, tol, 4 * k)
        dtw = time.time() - t0
        st = h.stats()
        ok = np.array_equal(np.nonzero(x)[0], sup)
        print("solve %d: iter=%d err=%.3e support_ok=%s wall=%.2f ms dev=%.2f ms sweeps=%d avg_sweep=%.4f ms (%.1f GB/s) maxcoeferr=%.2e" % (
            i, it, err, ok, dtw * 1e3, st["solve_ms"], st["sweep_launches"],
            st["sweep_ms"] / max(1, st["sweep_launches"]),
            st["sweep_bytes"] / (st["sweep_ms"] / max(1, st["sweep_launches"])) / 1e6,
            np.abs(x[sup] - coef).max()), flush=True)
    h.set_profiling(False)
    for la in (1, 2, 4, 8):
        h.set_option("lookahead", la)
        t0 = time.time()
        x, it, err = h.solve(y, tol, 4 * k)
        print("lookahead %d: wall %.2f ms iter %d" % (la, (time.time() - t0) * 1e3, it), flush=True)
    print(json.dumps({"variants_ms": out}))


if __name__ == "__main__":
    main()
