#!/usr/bin/env python3
"""GPU probe: configs[2] (4096 signals sharing the 8192 x 65536 matrix, Gram form) with the scan fused into the Gram-form
pass (k_la_cqs) and as two kernels (k_la_cq + k_scansel): seconds per batch, k_la_cq(s) HBM rate, records equal?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-solvers_amd", "python")]
import numpy as np, torch, sship
import bench
dev = torch.device("cuda:0")
A = torch.from_numpy(bench.survey_matrix()).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Y, sups, coefs = bench.make_batch(A, 4242, B, 64, torch)
h = sship.Homotopy(A)
rb = h.record_bytes(96)
rec = torch.zeros((B, rb), dtype=torch.uint8, device=dev)
h.solve_batch_compact(Y, 1e-3, 256, kmax=96, out=rec)      # G
torch.cuda.synchronize()
keep = {}
vec = 0
for fused, cols, rows in ((1, 4, 4), (1, 8, 2), (1, 16, 1), (1, 16, 2), (1, 32, 1), (0, 4, 4)):
    h.set_option("batch_fused_scan", fused)
    h.set_option("cq_cols", cols)
    h.set_option("cq_rows", rows)
    h.set_profiling(True)
    h.reset_stats()
    t0 = time.perf_counter()
    h.solve_batch_compact(Y, 1e-3, 256, kmax=96, out=rec)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = h.stats()
    keep[(fused, cols, rows)] = rec.cpu().numpy().copy()
    print("cols %d rows %d fused %d: %.4f s = %.0f signals/s; pass kernel %.1f ms over %d launches = %.3f ms each, %.0f GB/s; tie_reruns %d" % (
        cols, rows, fused, dt, B / dt, st["cq_ms"], st["cq_launches"], st["cq_ms"] / max(1, st["cq_launches"]),
        st["cq_bytes"] / max(1e-9, st["cq_ms"] * 1e-3) / 1e9, st["tie_reruns"]), flush=True)
    h.set_profiling(False)
    h.reset_stats()
    t0 = time.perf_counter()
    h.solve_batch_compact(Y, 1e-3, 256, kmax=96, out=rec)
    torch.cuda.synchronize()
    print("          unprofiled: %.4f s = %.0f signals/s" % (time.perf_counter() - t0, B / (time.perf_counter() - t0)), flush=True)
print("records equal:", all(np.array_equal(v, list(keep.values())[0]) for v in keep.values()))
