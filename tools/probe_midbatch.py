"""Mid-size batches at configs[1] size (8192 x 65536 fp32, k = 64): column form against one solve per signal
(batch_cols_min = 0) and, for B >= 192, the two-GEMM form."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, ROOT)
import torch
import sship
from bench import survey_matrix

m, n, k = 8192, 65536, 64
A = survey_matrix()
rng = np.random.default_rng(99)
Bmax = 256
X0 = np.zeros((Bmax, n), np.float32)
for b in range(Bmax):
    X0[b, rng.choice(n, k, replace=False)] = 1 + np.abs(rng.standard_normal(k))
Ad = torch.from_numpy(A).to("cuda:0")
Yd = (torch.from_numpy(X0).to("cuda:0") @ Ad.t()).contiguous()
del X0
with sship.Homotopy(Ad) as h:
    del Ad
    for B in (16, 24, 32, 64, 128, 191, 256):
        out = torch.zeros((B, n), dtype=torch.float32, device="cuda:0")
        res = {}
        for label, cmin in (("cols", 24 if B >= 24 else B), ("old", 0)):
            h.set_option("batch_cols_min", cmin)
            h.solve_batch(Yd[:B], 1e-3, 96, out=out)          # warm-up (allocations)
            torch.cuda.synchronize()
            h.reset_stats()
            t0 = time.perf_counter()
            _, iters, errs = h.solve_batch(Yd[:B], 1e-3, 96, out=out)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st = h.stats()
            res[label] = (dt, out.clone(), iters.copy())
            print("B %4d %-5s %8.2f ms  %7.1f signals/s  iters %d..%d  col rounds %d  lock-step rounds %d" % (
                B, label, dt * 1e3, B / dt, iters.min(), iters.max(), st["batch_col_rounds"], st["batch_rounds"]), flush=True)
        a, b_ = res["cols"][1], res["old"][1]
        print("   same iterations:", bool(np.array_equal(res["cols"][2], res["old"][2])),
              " same supports:", bool(torch.equal(a != 0, b_ != 0)),
              " max |dx| / max |x|: %.2e" % float((a - b_).abs().max() / b_.abs().max()), flush=True)
