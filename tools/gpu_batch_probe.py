#!/usr/bin/env python3
"""GPU-box probe of BASELINE.json configs[2]: a batch of signals sharing one 8192x65536 fp32
sensing matrix, solved in lock-step (MFMA GEMM correlations)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import torch
import sship
ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="64,512,4096")
ap.add_argument("--k", type=int, default=64)
ap.add_argument("--max-iter", type=int, default=256)
args = ap.parse_args()
m, n, k = 8192, 65536, args.k
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32) / np.sqrt(m)
h = sship.Homotopy(A)
for B in [int(b) for b in args.batches.split(",")]:
    rng = np.random.default_rng(99)
    sup = np.stack([np.sort(rng.choice(n, k, replace=False)) for _ in range(B)])
    coef = 1.0 + np.abs(rng.standard_normal((B, k)))
    Y = torch.empty((B, m), device=dev, dtype=torch.float32)
    for b0 in range(0, B, 256):
        b1 = min(B, b0 + 256)
        idx = torch.from_numpy(sup[b0:b1]).to(dev)                      # (bb, k)
        cols = A.t()[idx.reshape(-1)].reshape(b1 - b0, k, m).double()   # (bb, k, m)
        Y[b0:b1] = torch.einsum("bkm,bk->bm", cols, torch.from_numpy(coef[b0:b1]).to(dev)).float()
    X = torch.zeros((B, n), device=dev, dtype=torch.float32)
    torch.cuda.synchronize()
    h.reset_stats()
    t0 = time.perf_counter()
    _, iters, errs = h.solve_batch(Y, 1e-3, args.max_iter, out=X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = h.stats()
    nz = (X != 0)
    ok = 0
    Xh_rows = X.cpu().numpy() if B <= 512 else None
    for b in range(B):
        row = Xh_rows[b] if Xh_rows is not None else X[b].cpu().numpy()
        if np.array_equal(np.nonzero(row)[0], sup[b]) and np.abs(row[sup[b]] - coef[b]).max() < 1e-4 * coef[b].max():
            ok += 1
    rounds = st["batch_rounds"]
    print("   dbg: ndone", h.get_option("dbg_ndone"), "tile skip flags set", h.get_option("dbg_skip_sum"), "of", (B + 127) // 128)
    print("B=%5d: %.3f s  %.1f signals/s  rounds=%d (%.2f ms/round, GEMM flops %.1f TFLOP/s incl. tails)  iters min/mean/max %d/%.1f/%d  recovered %d/%d" % (
        B, dt, B / dt, rounds, dt / max(1, rounds) * 1e3,
        (rounds * 2 + 1) * 2.0 * ((B + 127) // 128 * 128) * n * m / dt / 1e12,
        iters.min(), iters.mean(), iters.max(), ok, B), flush=True)
    if B == int(args.batches.split(",")[-1]):
        bad = [b for b in range(B) if iters[b] >= 200]
        print("stragglers:", bad[:8])
        h.set_option("trace", 1)
        for b in bad[:3]:
            y = Y[b].contiguous()
            x1, it1, e1 = h.solve(y, 1e-3, 256)
            tr = h.trace()
            c0 = (A.t().double() @ y.double())
            j0 = int(torch.argmax(c0.abs()).item())
            print("signal", b, "single-path iters", it1, "err %.3e" % e1, "lead corr", float(c0[j0]), "planted?", j0 in set(sup[b].tolist()))
            print("   trace idx", tr["idx"][:12].tolist(), "added", tr["added"][:12].tolist())
            print("   gamma", np.array2string(tr["gamma"][:12], precision=4))
            rem = int((tr["added"] == 0).sum())
            print("   removals", rem, "distinct cols", len(set(tr["idx"].tolist())), "not in planted", len(set(tr["idx"].tolist()) - set(sup[b].tolist())))
