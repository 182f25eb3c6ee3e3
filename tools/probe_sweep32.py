#!/usr/bin/env python3
"""GPU probe: lookahead sweep (k_gemm32_tn_f32) variants at the C2 size: time, GB/s, correctness."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
import torch
m, n = 8192, 65536
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.randn(m, n, device="cuda", generator=g, dtype=torch.float32) / np.sqrt(m))
cols = np.arange(0, 32 * 1000, 1000, dtype=np.uint32)
ref = None
with sship.Homotopy(A) as h:
    bytes_ = m * n * 4 + 32 * m * 4 + 32 * n * 4
    for v in (0, 6, 7, 0, 6, 7, 0, 6, 7):
        h.set_option("sweep32_variant", v)
        G, ms = h.gram_cols(cols, 20)
        if ref is None:
            ref = (A.T @ A[:, torch.from_numpy(cols.astype(np.int64)).cuda()]).T.cpu().numpy()
        err = np.abs(G - ref).max()
        print("variant %d: %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)  max err %.2e" % (v, ms, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80, err), flush=True)
