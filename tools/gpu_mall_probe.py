import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import torch, sship
m, n = 8192, 65536
g = torch.Generator(device="cuda:0").manual_seed(1234)
A = torch.randn((m, n), generator=g, device="cuda:0", dtype=torch.float32) / np.sqrt(m)
h = sship.Homotopy(A)
r = np.random.default_rng(0).standard_normal(m).astype(np.float32)
c0, _ = h.gemv_t(r)
for tc in (0, 1024, 2048, 4096, 5120, 6144, 7168, 8192, 16384, 65536):
    h.set_option("temporal_cols", tc)
    h.gemv_t(r, 5)
    best = min(h.gemv_t(r, 40)[1] for _ in range(3))
    c, _ = h.gemv_t(r)
    print("temporal_cols %6d (%4d MB): %.4f ms  %.0f GB/s  same=%s" % (tc, tc * 32 // 1024, best, (m * n * 4 + m * 4 + n * 4) / best / 1e6, np.array_equal(c, c0)), flush=True)
