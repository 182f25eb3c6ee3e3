"""developer probe: the early form on a wide dictionary (passes dealt out by shader engine, any tile count)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "sparse-solvers_amd", "python")]
import numpy as np, torch, sship
n = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
m, k = 8192, 64
g = torch.Generator(device="cuda:0").manual_seed(7)
A = torch.randn((m, n), generator=g, device="cuda:0", dtype=torch.float32) / np.sqrt(m)
rng = np.random.default_rng(3)
sigs = []
for s in range(6):
    sup = np.sort(rng.choice(n, k, replace=False))
    coef = torch.from_numpy((1 + np.abs(rng.standard_normal(k))).astype(np.float32)).to("cuda:0")
    sigs.append((A[:, torch.from_numpy(sup).to("cuda:0")] @ coef, sup))
h = sship.Homotopy(A)
x = torch.zeros(n, device="cuda:0")
ref = {}
for se in (0, 1, 1, 1):
    h.set_option("early_se", se)
    for i, (y, sup) in enumerate(sigs):
        h.reset_stats()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, it, e = h.solve(y, 1e-3, 256, out=x)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = h.stats()
        xs = x.cpu().numpy()
        if se == 0: ref[i] = xs.copy()
        print("early_se %d signal %d: %.3f ms iter %d solo %d retries %d sweeps %d same-as-se0 %s support ok %s" % (
            se, i, dt * 1e3, it, st["solo_solves"], st["solo_retries"], st["lookahead_sweeps"], np.array_equal(xs, ref[i]),
            np.array_equal(np.nonzero(xs)[0], sup)), flush=True)
