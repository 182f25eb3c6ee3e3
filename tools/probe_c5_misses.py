"""Developer probe: configs[4] (fp64), which iterations the lookahead passes fall on (run under rocprofv3 --kernel-trace
and count the k_la_iter launches between the passes), for three signals."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, torch, sship
m5, n5, k5 = 16384, 131072, 128
g5 = torch.Generator(device="cuda:0").manual_seed(4321)
A5 = torch.randn((m5, n5), generator=g5, device="cuda:0", dtype=torch.float64)
A5 /= np.sqrt(m5)
h = sship.Homotopy(A5)
x = torch.zeros(n5, device="cuda:0", dtype=torch.float64)
import time
res = {}
for late in (64, 32):
    h.set_option("sweep_cols_f64_late", late)
    for s in range(6):
        rng = np.random.default_rng(4322 + s)
        sup = np.sort(rng.choice(n5, k5, replace=False))
        coef = 1.0 + np.abs(rng.standard_normal(k5))
        y = (A5[:, torch.from_numpy(sup).cuda()] @ torch.from_numpy(coef).cuda()).contiguous()
        h.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, it, e = h.solve(y, 1e-9, 512, out=x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        xs = x.cpu().numpy()
        ok = np.array_equal(np.nonzero(xs)[0], sup)
        same = res.setdefault(s, xs.copy()) is xs or np.array_equal(res[s], xs)
        print("late cols", late, "signal", s, "iters", it, "passes after A^T y", h.stats()["lookahead_sweeps"], "%.2f ms" % (dt * 1e3),
              "support ok", ok, "same bits as the first setting", bool(np.array_equal(res[s], xs)))
