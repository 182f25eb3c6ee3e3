import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import torch, sship
m, n = 8192, 65536
g = torch.Generator(device="cuda:0").manual_seed(1234)
A = torch.randn((m, n), generator=g, device="cuda:0", dtype=torch.float32) / np.sqrt(m)
h = sship.Homotopy(A)
rng = np.random.default_rng(3)
for S in (1, 7, 32):
    cols = rng.choice(n, S, replace=False).astype(np.uint32)
    G, ms = h.gram_cols(cols, 3)
    G, ms = h.gram_cols(cols, 20)
    ref = (A[:, torch.from_numpy(cols.astype(np.int64)).to("cuda:0")].double().T @ A.double()).cpu().numpy()
    err = np.abs(G - ref).max() / np.abs(ref).max()
    by = m * n * 4 + S * m * 4 + S * n * 4
    print("S=%2d: %.4f ms  %.0f GB/s  %.1f TFLOP/s(32 rows)  rel err %.2e" % (S, ms, by / ms / 1e6, 2.0 * 32 * m * n / ms / 1e9, err), flush=True)
c, ms1 = h.gemv_t(A[:, int(cols[0])].contiguous().cpu().numpy(), 10)
print("vs 1-rhs sweep: %.4f ms, max diff to gram row: %.2e" % (ms1, np.abs(c - G[0]).max()))
