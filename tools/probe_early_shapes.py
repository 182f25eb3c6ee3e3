"""Developer probe: the early form with / without the passes dealt out by shader engine (early_se 1 / 0) at dictionary widths
between 24 000 and 131 072."""
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
        for ep in (1, 0, 3, 1, 0):
            h.set_option("early_se", ep)
            h.solve(sigs[0], 1e-3, 256, out=x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for y in sigs[2:]:
                h.solve(y, 1e-3, 256, out=x)
            torch.cuda.synchronize()
            out.append("%d: %.3f" % (ep, (time.perf_counter() - t0) / len(sigs[2:]) * 1e3))
        print("n = %6d  ms/solve by early_se  %s" % (N, "  ".join(out)), flush=True)
    torch.cuda.empty_cache()
