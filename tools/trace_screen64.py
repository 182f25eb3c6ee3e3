#!/usr/bin/env python3
"""configs[4] solves in the fp64 screened form, for `rocprofv3 --kernel-trace --stats -- python3 tools/trace_screen64.py`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship  # noqa: E402

dev = torch.device("cuda", 0)
m5, n5, k5 = 16384, 131072, 128
g5 = torch.Generator(device=dev).manual_seed(4321)
A5 = torch.randn((m5, n5), generator=g5, device=dev, dtype=torch.float64)
A5 /= np.sqrt(m5)
rng5 = np.random.default_rng(4322)
sup5 = np.sort(rng5.choice(n5, k5, replace=False))
coef5 = 1.0 + np.abs(rng5.standard_normal(k5))
y5 = (A5[:, torch.from_numpy(sup5).to(dev)] @ torch.from_numpy(coef5).to(dev)).contiguous()
h5 = sship.Homotopy(A5, device=0)
del A5
torch.cuda.empty_cache()
if len(sys.argv) > 1:
    h5.set_option("screen_single", int(sys.argv[1]))
x5 = torch.zeros(n5, device=dev, dtype=torch.float64)
for _ in range(4):
    torch.cuda.synchronize()
    _, it, e = h5.solve(y5, 1e-9, 512, out=x5)
print(it, h5.stats()["screen_signals"])
h5.close()
