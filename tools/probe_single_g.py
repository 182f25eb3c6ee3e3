"""Single-signal solves with G = A^T A in HBM (the subset form for one signal): ms per solve."""
import sys, time
import numpy as np
sys.path.insert(0, "sparse-solvers_amd/python"); sys.path.insert(0, ".")
import torch, sship
import bench as B
dev = torch.device("cuda:0")
A = torch.from_numpy(B.survey_matrix()).to(dev)
sigs = [B.make_signal(A, 1235 + s, B.K_SPARSE, torch) for s in range(12)]
x = torch.zeros(B.N, device=dev)
with sship.Homotopy(A) as h:
    h.set_option("gram_full_after", 1)
    h.solve(sigs[0][0], 1e-3, 256, out=x); h.solve(sigs[1][0], 1e-3, 256, out=x); torch.cuda.synchronize()
    h.reset_stats()
    t = time.perf_counter(); ok = 0
    for s in range(2, 12):
        h.solve(sigs[s][0], 1e-3, 256, out=x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print("with G: ms/solve", round(dt * 1e3, 4), "accepted", int(h.stats()["subset_signals"]), "redone", int(h.stats()["subset_redone"]))
