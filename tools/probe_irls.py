#!/usr/bin/env python3
"""IRLS on the device: construction (Householder QR, Q, Q^T Q) and the Newton loop, blocked form against the one-workgroup form
(SS_HIP_IRLS_FUSED=1), at 4096 x 1024 and 2048 x 512 fp32 / fp64."""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
if len(sys.argv) > 1 and sys.argv[1] == "both":
    for fused in ("", "1"):
        env = dict(os.environ)
        if fused:
            env["SS_HIP_IRLS_FUSED"] = "1"
        else:
            env.pop("SS_HIP_IRLS_FUSED", None)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=True)
    sys.exit(0)
import torch
import sship
dev = torch.device("cuda", 0)
for (mi, ni, dt) in [(4096, 1024, np.float32), (2048, 512, np.float32), (2048, 512, np.float64)]:
    rng = np.random.default_rng(777)
    A = (rng.standard_normal((mi, ni)) / np.sqrt(mi)).astype(dt)
    x0 = np.zeros(ni, dt)
    x0[rng.choice(ni, 8, replace=False)] = (1.0 + np.abs(rng.standard_normal(8))).astype(dt)
    y = (A.astype(np.float64) @ x0).astype(dt)
    Ad = torch.from_numpy(A).to(dev)
    yd = torch.from_numpy(y).to(dev)
    h = sship.Irls(Ad, device=0)
    h.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = sship.Irls(Ad, device=0)
    torch.cuda.synchronize()
    tc = time.perf_counter() - t0
    h.solve(yd, 1e-3, 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        x, it, e, spd = h.solve(yd, 1e-3, 8)
    torch.cuda.synchronize()
    ts = (time.perf_counter() - t0) / 3
    xh = x.cpu().numpy() if hasattr(x, "cpu") else x
    print("%s IRLS %d x %d %s: construct %.2f ms, solve %.2f ms (%d iterations, %.2f ms each), spd failure %d, support %s"
          % ("one-workgroup" if os.environ.get("SS_HIP_IRLS_FUSED") else "blocked      ", mi, ni, np.dtype(dt).name, tc * 1e3, ts * 1e3, it, ts * 1e3 / max(1, it), spd,
             np.array_equal(np.nonzero(xh > 0.01)[0], np.nonzero(x0)[0])), flush=True)
    h.close()
