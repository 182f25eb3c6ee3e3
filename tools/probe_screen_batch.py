#!/usr/bin/env python3
"""Batches without G in the screened form (option batch_screen) at configs[1] size: B = 8, 64, 256 against the column form /
one solve per signal, planted supports, agreement with single solves."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
M, N, K = 8192, 65536, 64
dev = torch.device("cuda", 0)
A_host = np.random.default_rng(1234).standard_normal((M, N), dtype=np.float32)
A_host /= np.float32(np.sqrt(M))
A = torch.from_numpy(A_host).to(dev)
del A_host
def make(B, seed):
    rng = np.random.default_rng(seed)
    sups = np.stack([np.sort(rng.choice(N, K, replace=False)) for _ in range(B)])
    coefs = 1.0 + np.abs(rng.standard_normal((B, K)))
    Y = torch.empty((B, M), device=dev, dtype=torch.float32)
    for b in range(B):
        Y[b] = (A[:, torch.from_numpy(sups[b]).to(dev)].double() @ torch.from_numpy(coefs[b]).to(dev)).float()
    return Y.contiguous(), sups, coefs
for B in (8, 64, 256):
    Y, sups, coefs = make(B, 4000 + B)
    for mode in (1, 0):
        with sship.Homotopy(A, device=0) as h:
            h.set_option("batch_screen", mode)
            h.set_option("batch_gram_min", 100000)            # (no G in this probe)
            X = torch.zeros((B, N), device=dev)
            h.solve_batch(Y, 1e-3, 256, out=X)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                _, its, errs = h.solve_batch(Y, 1e-3, 256, out=X)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            st = h.stats()
            Xh = X.cpu().numpy()
            ok = sum(np.array_equal(np.nonzero(Xh[b])[0], sups[b]) for b in range(B))
            cerr = max(np.abs(Xh[b][sups[b]] - coefs[b]).max() / coefs[b].max() for b in range(B))
            print("B %4d batch_screen %d: %.2f ms = %.0f signals/s, supports exact %d / %d, max rel coef err %.2e, iterations %d..%d, screened %d redone %d, column rounds %d"
                  % (B, mode, dt * 1e3, B / dt, ok, B, cerr, int(np.min(its)), int(np.max(its)), st["screen_signals"], st["screen_redone"], st["batch_col_rounds"]), flush=True)
