"""Developer probe: what a single-signal solve does at wide dictionaries (stats of the speculative form)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
M, K = 8192, 64
for N in [int(v) for v in os.environ.get("PROBE_N", "131072,163840,196608,262144").split(",")]:
    g = torch.Generator(device="cuda:0").manual_seed(N)
    Ad = torch.randn((M, N), generator=g, device="cuda:0", dtype=torch.float32) / np.sqrt(M)
    sigs = []
    for s in range(10):
        rng = np.random.default_rng(99 + s)
        sup = np.sort(rng.choice(N, K, replace=False))
        coef = 1.0 + np.abs(rng.standard_normal(K))
        sigs.append((Ad[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).cuda()).float().contiguous())
    x = torch.zeros(N, device="cuda:0")
    with sship.Homotopy(Ad) as h:
        del Ad
        h.solve(sigs[0], 1e-3, 256, out=x); torch.cuda.synchronize()
        h.reset_stats(); h.set_profiling(True)
        t0 = time.perf_counter()
        its = []
        for y in sigs[2:]:
            _, it, _ = h.solve(y, 1e-3, 256, out=x); its.append(it)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(sigs[2:]) * 1e3
        st = h.stats()
        keys = ("solves", "iterations", "lookahead_sweeps", "solo_solves", "solo_retries", "sweep32_launches", "sweep32_ms", "sweep1_ms", "sweep1_launches", "persist_fallbacks", "gram_fallbacks", "tie_reruns")
        print("n = %6d  %.3f ms/solve  iters %s  %s" % (N, dt, its[:3], {k: (round(st[k], 3) if isinstance(st[k], float) else int(st[k])) for k in keys}), flush=True)
    torch.cuda.empty_cache()
