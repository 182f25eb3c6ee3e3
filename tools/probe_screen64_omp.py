#!/usr/bin/env python3
"""OMP through the fp64 screened form: small forced shapes against the default OMP engine, then configs[4] timed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship  # noqa: E402

for (m, n, k, seed) in [(1024, 16384, 24, 1), (1536, 9000, 40, 3), (2048, 16384, 60, 2)]:
    rng = np.random.default_rng(5000 + seed)
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    sup = np.sort(rng.choice(n, k, replace=False))
    x0 = np.zeros(n)
    x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
    y = A @ x0
    with sship.Homotopy(A, device=0) as h:
        h.set_option("screen_single", 2)
        x, it, err = h.solve_omp(y, 1e-9, 4 * k)
        st = h.stats()
        h.set_option("screen_single", 0)
        xd, itd, errd = h.solve_omp(y, 1e-9, 4 * k)
    print("OMP m %5d n %6d k %3d | screened %d redone %d headroom %.3f | picks %d / default %d | support %s | max |x - x_default| / max |x| %.2e | err %.2e"
          % (m, n, k, st["screen_signals"], st["screen_redone"], st["screen_headroom"], it, itd,
             np.array_equal(np.nonzero(x)[0], sup), np.abs(x - xd).max() / np.abs(xd).max(), np.abs(x[sup] - x0[sup]).max() / x0.max()), flush=True)

if "--no-big" not in sys.argv:
    import torch
    dev = torch.device("cuda", 0)
    m5, n5, k5 = 16384, 131072, 128
    g5 = torch.Generator(device=dev).manual_seed(4321)
    A5 = torch.randn((m5, n5), generator=g5, device=dev, dtype=torch.float64)
    A5 /= np.sqrt(m5)
    rng5 = np.random.default_rng(4322)
    sup5 = np.sort(rng5.choice(n5, k5, replace=False))
    coef5 = 1.0 + np.abs(rng5.standard_normal(k5))
    y5 = (A5[:, torch.from_numpy(sup5).to(dev)] @ torch.from_numpy(coef5).to(dev)).contiguous()
    h5 = sship.Homotopy(A5, device=0)
    del A5
    torch.cuda.empty_cache()
    x5 = torch.zeros(n5, device=dev, dtype=torch.float64)
    for mode in (1, 0):
        h5.set_option("screen_single", mode)
        h5.reset_stats()
        h5.solve_omp(y5, 1e-9, 512, out=x5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            _, it, e = h5.solve_omp(y5, 1e-9, 512, out=x5)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        xh = x5.cpu().numpy()
        st = h5.stats()
        print("configs[4] OMP screen_single %d: %.3f ms per solve, picks %d, support exact %s, max rel coef err %.2e, screened %d redone %d headroom %.3f"
              % (mode, dt * 1e3, it, np.array_equal(np.nonzero(xh)[0], sup5), np.abs(xh[sup5] - coef5).max() / coef5.max(),
                 st["screen_signals"], st["screen_redone"], st["screen_headroom"]), flush=True)
    h5.close()
