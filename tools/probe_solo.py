#!/usr/bin/env python3
"""GPU probe: C2-size solve time in the resident (la_fused 2) and speculative (la_fused 3) forms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import numpy as np, torch, sship
m, n, k = 8192, 65536, 64
g = torch.Generator(device="cuda").manual_seed(1234)
A = torch.randn((m, n), generator=g, device="cuda", dtype=torch.float32) / np.sqrt(m)
sigs = []
for s in range(8):
    rng = np.random.default_rng(1235 + s)
    sup = np.sort(rng.choice(n, k, replace=False)); coef = 1.0 + np.abs(rng.standard_normal(k))
    sigs.append((A[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).cuda()).float())
with sship.Homotopy(A) as h:
    ref = None
    for mode in (2, 3, 2, 3):
        h.set_option("la_fused", mode)
        h.reset_stats()
        for y in sigs[:2]:
            h.solve(y, 1e-3, 256)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        xs = [h.solve(y, 1e-3, 256) for y in sigs]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(sigs)
        st = h.stats()
        print("la_fused %d: %.3f ms per solve, iterations %s, solo solves %d, failed checks %d" % (
            mode, dt * 1e3, sorted(set(int(x[1]) for x in xs)), st["solo_solves"], st["solo_retries"]), flush=True)
        if ref is None:
            ref = xs
        else:
            assert all(np.array_equal(a[0], b[0]) and a[1] == b[1] for a, b in zip(ref, xs)), "forms disagree"
print("all forms bit-identical")
