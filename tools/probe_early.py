"""Developer probe: the early form at C2 — solve time and the event-timed pass on the second stream, with and
without the overlap (option early_probe = 1 runs the passes first, then the solo launch)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
M, N, K = 8192, 65536, 64
A = np.random.default_rng(1234).standard_normal((M, N), dtype=np.float32)
A /= np.float32(np.sqrt(M))
Ad = torch.from_numpy(A).to("cuda:0")
sigs = []
for s in range(42):
    rng = np.random.default_rng(1235 + s)
    sup = np.sort(rng.choice(N, K, replace=False))
    coef = 1.0 + np.abs(rng.standard_normal(K))
    y = (Ad[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).cuda()).float().contiguous()
    sigs.append(y)
x = torch.zeros(N, device="cuda:0")
with sship.Homotopy(Ad) as h:
    for name, opts in (("plain", {"early_solo": 0}),
                       ("early e-kernel |c0|", {"early_solo": 1, "early_probe": 0, "early_pass": 0, "early_adapt": 0}),
                       ("early LDSx3 |c0|", {"early_solo": 1, "early_probe": 0, "early_pass": 2, "early_adapt": 0, "early_se": 0}),
                       ("early LDSx3 adaptive", {"early_solo": 1, "early_probe": 0, "early_pass": 2, "early_adapt": 1, "early_se": 0}),
                       ("early by SE adaptive", {"early_solo": 1, "early_probe": 0, "early_pass": 2, "early_adapt": 1, "early_se": 1}),
                       ("early, no overlap", {"early_solo": 1, "early_probe": 1, "early_pass": 2, "early_adapt": 1})):
        for k_, v_ in opts.items():
            h.set_option(k_, v_)
        for prof in (0, 1):
            h.set_profiling(bool(prof))
            h.solve(sigs[0], 1e-3, 256, out=x)
            h.reset_stats()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for y in sigs[2:]:
                h.solve(y, 1e-3, 256, out=x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / len(sigs[2:])
            st = h.stats()
            print("%-20s profiling %d: %.3f ms/solve, sweeps/solve %.2f, timed 32-col pass %.3f ms (%d), A^T y %.3f ms, solo retries %d" % (
                name, prof, dt * 1e3, st["lookahead_sweeps"] / st["solves"], st["sweep32_ms"] / max(1, st["sweep32_launches"]),
                st["sweep32_launches"], st["sweep1_ms"] / max(1, st["sweep1_launches"]), st["solo_retries"]))
