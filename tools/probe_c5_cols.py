"""Developer probe: configs[4] (fp64, 16384 x 131072, k = 128) with 32 / 64 columns per lookahead sweep."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, torch, sship
m5, n5, k5 = 16384, 131072, 128
g5 = torch.Generator(device="cuda:0").manual_seed(4321)
A5 = torch.randn((m5, n5), generator=g5, device="cuda:0", dtype=torch.float64)
A5 /= np.sqrt(m5)
rng = np.random.default_rng(4322)
sup = np.sort(rng.choice(n5, k5, replace=False))
coef = 1.0 + np.abs(rng.standard_normal(k5))
y = (A5[:, torch.from_numpy(sup).cuda()] @ torch.from_numpy(coef).cuda()).contiguous()
h = sship.Homotopy(A5)
del A5
torch.cuda.empty_cache()
x = torch.zeros(n5, device="cuda:0", dtype=torch.float64)
res = {}
for cols, variant in ((32, 0), (32, 1), (32, 2), (64, 0)):
    h.set_option("sweep_cols_f64", cols)
    h.set_option("sweep_f64_variant", variant)
    h.set_profiling(True)
    h.solve(y, 1e-9, 512, out=x)
    h.reset_stats()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        _, it, e = h.solve(y, 1e-9, 512, out=x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    st = h.stats()
    xs = x.cpu().numpy()
    res[(cols, variant)] = xs.copy()
    print("sweep cols", cols, "variant", variant, "pass ms %.3f" % (st["sweep32_ms"] / max(1, st["sweep32_launches"])), "ms/solve %.2f" % (dt * 1e3), "iters", it, "sweeps/solve", st["lookahead_sweeps"] / st["solves"],
          "support ok", np.array_equal(np.nonzero(xs)[0], sup), "coef err", np.abs(xs[sup] - coef).max() / coef.max())
print("bitwise equal across tilings:", all(np.array_equal(res[(32, 0)], v) for v in res.values()))
