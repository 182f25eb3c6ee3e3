#!/usr/bin/env python3
"""GPU-box probe: correctness and TFLOP/s of the batched-correlation MFMA GEMM."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import torch
import sship
m, n = 8192, 65536
g = torch.Generator(device="cuda:0").manual_seed(1234)
A = torch.randn((m, n), generator=g, device="cuda:0", dtype=torch.float32) / np.sqrt(m)
h = sship.Homotopy(A)
for B in (1, 5, 128, 1024, 4096):
    R = torch.randn((B, m), generator=g, device="cuda:0", dtype=torch.float32)
    C = torch.empty((B, n), device="cuda:0", dtype=torch.float32)
    _, ms = h.gemm_t(R, repeats=1, out=C)
    _, ms = h.gemm_t(R, repeats=3, out=C)
    torch.cuda.synchronize()
    rows = [0, B // 2, B - 1]
    ref = (R[rows].double() @ A.double())
    err = (C[rows].double() - ref).abs().max().item() / ref.abs().max().item()
    Bp = (B + 127) // 128 * 128
    print("B=%5d  %.3f ms  %.1f TFLOP/s (padded rows %d)  %.1f TFLOP/s useful  rel err %.2e" % (
        B, ms, 2.0 * Bp * n * m / ms / 1e9, Bp, 2.0 * B * n * m / ms / 1e9, err), flush=True)
# against the GEMV path
r = R[0].contiguous()
c1, _ = h.gemv_t(r.cpu().numpy())
print("gemm vs gemv max diff:", float(np.abs(C[0].cpu().numpy() - c1).max()))
