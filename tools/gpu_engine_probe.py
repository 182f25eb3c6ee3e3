import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import torch, sship
m, n, k = 8192, 65536, 64
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1234)
A = torch.randn((m, n), generator=g, device=dev, dtype=torch.float32) / np.sqrt(m)
h = sship.Homotopy(A)
xs = {}
for eng in (0, 1):
    h.set_option("engine", eng)
    ts = []; oks = 0
    for s in range(8):
        rng = np.random.default_rng(1235 + s)
        sup = np.sort(rng.choice(n, k, replace=False)); coef = 1 + np.abs(rng.standard_normal(k))
        y = (A[:, torch.from_numpy(sup).to(dev)].double() @ torch.from_numpy(coef).to(dev)).float().contiguous()
        xd = torch.zeros(n, device=dev)
        h.reset_stats()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, it, err = h.solve(y, 1e-3, 256, out=xd)
        ts.append((time.perf_counter() - t0) * 1e3)
        x = xd.cpu().numpy()
        oks += int(np.array_equal(np.nonzero(x)[0], sup) and np.abs(x[sup] - coef).max() < 1e-4)
        xs[(eng, s)] = (x, it, err)
        sw = h.stats()["lookahead_sweeps"]
    print("engine %d: ms per solve %s  recovered %d/8  iters %d  lookahead sweeps(last) %d" % (eng, np.round(ts, 2), oks, it, sw), flush=True)
for s in range(8):
    a, b = xs[(0, s)], xs[(1, s)]
    print("signal %d: iters %d/%d  max|dx| %.2e  err %.2e/%.2e  support equal %s" % (s, a[1], b[1], np.abs(a[0] - b[0]).max(), a[2], b[2], np.array_equal(np.nonzero(a[0])[0], np.nonzero(b[0])[0])))
