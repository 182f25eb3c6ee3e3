#!/usr/bin/env python3
"""GPU probe: stage timestamps of k_la_iter over one fp64 solve at the configs[4] shape (SS_HIP_LA_DEBUG)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    env = dict(os.environ, SS_HIP_LA_DEBUG="/tmp/la_dbg_c5.bin")
    subprocess.run([sys.executable, __file__, "run"], env=env, check=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "la_stages.py"), "/tmp/la_dbg_c5.bin"], check=True)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, torch, sship, time
m, n, k = 16384, 131072, 128
g = torch.Generator(device="cuda:0").manual_seed(4321)
A = torch.randn((m, n), generator=g, device="cuda:0", dtype=torch.float64)
A /= np.sqrt(m)
rng = np.random.default_rng(4322)
sup = np.sort(rng.choice(n, k, replace=False))
coef = 1.0 + np.abs(rng.standard_normal(k))
y = (A[:, torch.from_numpy(sup).to("cuda:0")] @ torch.from_numpy(coef).to("cuda:0")).contiguous()
with sship.Homotopy(A) as h:
    del A
    torch.cuda.empty_cache()
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, it, err = h.solve(y, 1e-9, 512)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("fp64 configs[4]: %.2f ms per solve, iters %d, exact %s" % (dt * 1e3, it, np.array_equal(np.nonzero(x)[0], sup)), flush=True)
