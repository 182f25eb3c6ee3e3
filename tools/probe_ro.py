"""A/B of the reference-order sweep's two forms (option ro_staged: direct 16-byte loads / staged through LDS) on configs[1]."""
import sys, time, json
import numpy as np
sys.path.insert(0, "sparse-solvers_amd/python")
import torch
import sship

M, N, K = 8192, 65536, 64
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(1234)
A = (torch.randn((M, N), generator=g, dtype=torch.float32) / np.sqrt(M)).to(dev)
rng = np.random.default_rng(7)
x0 = np.zeros(N, np.float32); sup = rng.choice(N, K, replace=False); x0[sup] = 1 + np.abs(rng.standard_normal(K))
y = (A @ torch.from_numpy(x0).to(dev)).contiguous()
out = []
with sship.Homotopy(A) as h:
    h.set_option("engine", 3)
    x = torch.zeros(N, device=dev)
    ref = None
    for staged in (0, 1):
        if True:
            h.set_option("ro_staged", staged)
            _, ms = h.gemv_t(y, 5)
            h.solve(y, 1e-3, 256, out=x); torch.cuda.synchronize()
            t = time.perf_counter(); _, it, _ = h.solve(y, 1e-3, 256, out=x); torch.cuda.synchronize()
            dt = time.perf_counter() - t
            xs = x.clone()
            if ref is None: ref = xs
            out.append(dict(staged=staged, sweep1_ms=ms, frac=M * N * 4 / ms / 1e6 / 8000, solve_ms=dt * 1e3, iters=int(it), same_bits=bool(torch.equal(xs, ref))))
            print(out[-1], flush=True)
json.dump(out, open("gpurun_out/probe_ro.json", "w"))

# lock-step slots: 8 signals in engine 3, 1 / 2 / 4 per pass over A
rng = np.random.default_rng(9)
Ys = []
for b in range(8):
    x0 = np.zeros(N, np.float32); sup = rng.choice(N, K, replace=False); x0[sup] = 1 + np.abs(rng.standard_normal(K))
    Ys.append((A @ torch.from_numpy(x0).to(dev)))
Y8 = torch.stack(Ys).contiguous()
X8 = torch.zeros((8, N), device=dev)
res = []
with sship.Homotopy(A) as h:
    h.set_option("engine", 3)
    ref8 = None
    for slots in (1, 2, 4):
        h.set_option("ro_slots", slots)
        h.solve_batch(Y8, 1e-3, 256, out=X8); torch.cuda.synchronize()
        t = time.perf_counter(); _, it, _ = h.solve_batch(Y8, 1e-3, 256, out=X8); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if ref8 is None: ref8 = X8.clone()
        res.append(dict(slots=slots, ms_for_8=dt * 1e3, ms_per_signal=dt * 1e3 / 8, iters=[int(v) for v in it], same_bits=bool(torch.equal(X8, ref8))))
        print(res[-1], flush=True)
json.dump(dict(single=out, slots=res), open("gpurun_out/probe_ro.json", "w"))
