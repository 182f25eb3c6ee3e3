#!/usr/bin/env python3
"""configs[4] (A 16384 x 131072 fp64, k = 128) in the shipped default — the fp64 screened form with its resident tier (csrc/resident.hip) —
timed, Homotopy and OMP; with --tier2 also the sub-dictionary tier alone (option screen_resident = 0).
    python tools/probe_res64.py [--tier2] [--solves N] [--k K]          (under rocprofv3: tools/trace_screen_kernels.sh tools/probe_res64.py)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship  # noqa: E402


def main():
    import torch
    dev = torch.device("cuda", 0)
    nsolve = int(sys.argv[sys.argv.index("--solves") + 1]) if "--solves" in sys.argv else 6
    m5, n5, k5 = 16384, 131072, (int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 128)
    g5 = torch.Generator(device=dev).manual_seed(4321)
    A5 = torch.randn((m5, n5), generator=g5, device=dev, dtype=torch.float64)
    A5 /= np.sqrt(m5)
    sigs = []
    for s in range(nsolve + 1):
        rng5 = np.random.default_rng(4322 + s)
        sup5 = np.sort(rng5.choice(n5, k5, replace=False))
        coef5 = 1.0 + np.abs(rng5.standard_normal(k5))
        y5 = (A5[:, torch.from_numpy(sup5).to(dev)] @ torch.from_numpy(coef5).to(dev)).contiguous()
        sigs.append((y5, sup5, coef5))
    h5 = sship.Homotopy(A5, device=0)
    del A5
    torch.cuda.empty_cache()
    x5 = torch.zeros(n5, device=dev, dtype=torch.float64)
    modes = [("default", 1)] + ([("sub-dictionary tier only", 0)] if "--tier2" in sys.argv else [])
    for name, resident in modes:
        h5.set_option("screen_resident", resident)
        for which in ("homotopy", "omp"):
            solve = h5.solve if which == "homotopy" else h5.solve_omp
            solve(sigs[0][0], 1e-9, 512, out=x5)
            h5.reset_stats()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(nsolve):
                solve(sigs[r + 1][0], 1e-9, 512, out=x5)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / nsolve
            xh = x5.cpu().numpy()
            sup5, coef5 = sigs[nsolve][1], sigs[nsolve][2]
            st = h5.stats()
            why = {k_: v for k_, v in st.items() if k_.startswith("why_") and v}
            print("configs[4] %s, %s: %.3f ms per solve; last: support exact %s, max rel coef err %.2e; certified %d (resident %d, tier2 %d), redone %d, "
                  "headroom %.3f %s" % (which, name, dt * 1e3, bool(np.array_equal(np.nonzero(xh)[0], sup5)),
                                        float(np.abs(xh[sup5] - coef5).max() / coef5.max()), st["screen_signals"], st["screen_resident"],
                                        st["screen_tier2"], st["screen_redone"], st["screen_headroom"], why), flush=True)
    if "--batch" in sys.argv:
        nbt = int(sys.argv[sys.argv.index("--batch") + 1])
        Yb = torch.stack([sigs[i % len(sigs)][0] for i in range(nbt)]).contiguous()
        Xb = torch.zeros((nbt, n5), device=dev, dtype=torch.float64)
        h5.set_option("screen_resident", 1)
        h5.solve_batch(Yb, 1e-9, 512, out=Xb)
        h5.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h5.solve_batch(Yb, 1e-9, 512, out=Xb)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
        st = h5.stats()
        okb = 0
        Xh = Xb.cpu().numpy()
        for i in range(nbt):
            okb += int(np.array_equal(np.nonzero(Xh[i])[0], sigs[i % len(sigs)][1]))
        t0 = time.perf_counter()
        for i in range(nbt):
            h5.solve(Yb[i], 1e-9, 512, out=x5)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t0
        print("configs[4] batch of %d fp64 signals: %.2f ms = %.3f ms per signal (one solve per signal: %.3f ms each — %.2f x); supports exact %d / %d; "
              "certified in the batch %d, handed to the tiers behind %d" % (nbt, dtb * 1e3, dtb * 1e3 / nbt, dt1 * 1e3 / nbt, dt1 / dtb, okb, nbt,
                                                                           st["screen_resident"], st["screen_tier2"]), flush=True)
    h5.close()


if __name__ == "__main__":
    main()
