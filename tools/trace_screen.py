#!/usr/bin/env python3
"""configs[1] solves in the screened form, for `rocprofv3 --kernel-trace --stats -- python3 tools/trace_screen.py`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship  # noqa: E402

M, N, K = 8192, 65536, 64
A_host = np.random.default_rng(1234).standard_normal((M, N), dtype=np.float32)
A_host /= np.float32(np.sqrt(M))
dev = torch.device("cuda", 0)
A = torch.from_numpy(A_host).to(dev)
h = sship.Homotopy(A, device=0)
if len(sys.argv) > 1:
    h.set_option("screen_single", int(sys.argv[1]))
X = torch.zeros(N, device=dev, dtype=torch.float32)
for s in range(12):
    rng = np.random.default_rng(1235 + s)
    sup = np.sort(rng.choice(N, K, replace=False))
    coef = 1.0 + np.abs(rng.standard_normal(K))
    y = (A[:, torch.from_numpy(sup).to(dev)].double() @ torch.from_numpy(coef).to(dev)).float().contiguous()
    torch.cuda.synchronize()
    _, it, err = h.solve(y, 1e-3, 256, out=X)
print(h.stats()["screen_signals"], it)
h.close()
