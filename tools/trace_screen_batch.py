import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
M, N, K, B = 8192, 65536, 64, 64
dev = torch.device("cuda", 0)
A_host = np.random.default_rng(1234).standard_normal((M, N), dtype=np.float32)
A_host /= np.float32(np.sqrt(M))
A = torch.from_numpy(A_host).to(dev)
rng = np.random.default_rng(4064)
Y = torch.empty((B, M), device=dev, dtype=torch.float32)
for b in range(B):
    sup = np.sort(rng.choice(N, K, replace=False))
    coef = 1.0 + np.abs(rng.standard_normal(K))
    Y[b] = (A[:, torch.from_numpy(sup).to(dev)].double() @ torch.from_numpy(coef).to(dev)).float()
h = sship.Homotopy(A, device=0)
h.set_option("batch_gram_min", 100000)
X = torch.zeros((B, N), device=dev)
for _ in range(4):
    h.solve_batch(Y, 1e-3, 256, out=X)
torch.cuda.synchronize()
print(h.stats()["screen_signals"])
h.close()
