"""The screened form across dictionary widths (m = 8192, k = 64, fp32): ms per solve with and without it, and the time the
bytes alone would take at the rates of configs[1] (the two passes over the fp16 copy at 6.5 and 5.9 TB/s)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
M, K = 8192, 64
for N in (17000, 24000, 32768, 49152, 65536, 98304, 131072, 196608, 262144):
    g = torch.Generator(device="cuda:0").manual_seed(N)
    Ad = torch.randn((M, N), generator=g, device="cuda:0", dtype=torch.float32) / np.sqrt(M)
    sigs = []
    for s in range(12):
        rng = np.random.default_rng(99 + s)
        sup = np.sort(rng.choice(N, K, replace=False))
        coef = 1.0 + np.abs(rng.standard_normal(K))
        sigs.append((Ad[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).cuda()).float().contiguous())
    x = torch.zeros(N, device="cuda:0")
    with sship.Homotopy(Ad) as h:
        del Ad
        out = []
        for mode in (1, 0):
            h.set_option("screen_single", mode)
            h.reset_stats()
            h.solve(sigs[0], 1e-3, 256, out=x)
            h.solve(sigs[1], 1e-3, 256, out=x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for y in sigs[2:]:
                h.solve(y, 1e-3, 256, out=x)
            torch.cuda.synchronize()
            st = h.stats()
            out.append("%s %.3f ms (certified %d, redone %d)" % ("screened" if mode else "default engine", (time.perf_counter() - t0) / len(sigs[2:]) * 1e3,
                                                              st["screen_signals"], st["screen_redone"]))
        floor = M * N * 2 / 6.5e9 + M * N * 2 / 5.9e9 + 0.38 + 0.07          # + 64 iterations + select / Gs / tail
        print("n = %6d  %s | %s | bytes + iterations: %.3f ms" % (N, out[0], out[1], floor), flush=True)
    torch.cuda.empty_cache()
