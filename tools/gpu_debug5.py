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
rng = np.random.default_rng(1235)
sup = np.sort(rng.choice(n, k, replace=False)); coef = 1 + np.abs(rng.standard_normal(k))
yd = (A[:, torch.from_numpy(sup).to(dev)].double() @ torch.from_numpy(coef).to(dev)).float().contiguous()
yh = yd.cpu().numpy()
xd = torch.zeros(n, device=dev); xh = np.zeros(n, np.float32)
def run(name, y, out, reps=12):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.solve(y, 1e-3, 256, out=out)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts[2:])
    print("%-34s min %.2f med %.2f max %.2f ms" % (name, ts.min(), np.median(ts), ts.max()), flush=True)
run("host y, host out", yh, xh)
run("device y, device out", yd, xd)
h.set_profiling(True)
run("device, profiling every sweep", yd, xd)
h.set_option("profile_every", 8)
run("device, profiling every 8th", yd, xd)
h.set_profiling(False)
for la in (2, 8, 32):
    h.set_option("lookahead", la)
    run("device, lookahead %d" % la, yd, xd)
print(os.sched_getaffinity(0).__len__(), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "no cpu.max")
