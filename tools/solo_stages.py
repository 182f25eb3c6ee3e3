#!/usr/bin/env python3
"""GPU probe: stage timestamps (SS_HIP_LA_DEBUG) of one C2-size solve in the resident (la_fused 2) and
speculative (la_fused 3) forms, decoded with tools/la_stages2.py."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else None
if mode is None:
    for md in ("2", "3"):
        env = dict(os.environ, SS_HIP_LA_DEBUG="/tmp/la_dbg_%s.bin" % md)
        subprocess.run([sys.executable, __file__, md], env=env, check=True)
        print("== la_fused", md, flush=True)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "la_stages2.py"), "/tmp/la_dbg_%s.bin" % md], check=True)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, torch, sship, time
m, n, k = 8192, 65536, 64
g = torch.Generator(device="cuda").manual_seed(1234)
A = torch.randn((m, n), generator=g, device="cuda", dtype=torch.float32) / np.sqrt(m)
rng = np.random.default_rng(1235)
sup = np.sort(rng.choice(n, k, replace=False)); coef = 1.0 + np.abs(rng.standard_normal(k))
y = (A[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).cuda()).float()
with sship.Homotopy(A) as h:
    h.set_option("la_fused", int(mode))
    for _ in range(3):
        x, it, e = h.solve(y, 1e-3, 256)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        x, it, e = h.solve(y, 1e-3, 256)
    torch.cuda.synchronize()
    print("la_fused %s: %.3f ms per solve, iter %d, stats %s" % (mode, (time.perf_counter() - t0) * 100, it, {k_: v for k_, v in h.stats().items() if "solo" in k_}), flush=True)
