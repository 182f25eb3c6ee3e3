"""stress_screen.py case 224 (fp64, signed, derailed path): where does the path on the sub-dictionary leave the full one?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
rng = np.random.default_rng(20261004)
for case in range(225):
    f64 = case % 3 == 2
    m = int(rng.choice([512, 768, 1024, 1536, 2048]))
    n = int(rng.choice([8192, 12000, 16384] if f64 else [2048, 4096, 8192, 16384]))
    k = int(rng.integers(4, max(5, m // 24)))
    signed = bool(rng.integers(0, 2))
    noise = float(rng.choice([0.0, 0.0, 1e-4, 1e-2]))
    fixes = bool(rng.integers(0, 2))
    A = rng.standard_normal((m, n))
    sup = np.sort(rng.choice(n, k, replace=False))
    z = rng.standard_normal(k)
    sg = rng.choice([-1.0, 1.0], k) if signed else None
    if noise:
        rng.standard_normal(m)
A = A / np.sqrt(m)
x0 = np.zeros(n); x0[sup] = 1.0 + np.abs(z)
if signed: x0[sup] *= sg
y = A @ x0
budget = 3 * k + 8
print(case, m, n, k, signed, noise, fixes, flush=True)
with sship.Homotopy(A, device=0) as h:
    h.set_option("screen_single", 0)
    h.set_option("trace", 1)
    xf, itf, ef = h.solve(y, 1e-9, budget)
    trf = h.trace()
    c0, _ = h.gemv_t(y)
order = np.argsort(-np.abs(c0), kind="stable")
sub = np.sort(order[:2048])
with sship.Homotopy(np.ascontiguousarray(A[:, sub]), device=0) as hs:
    hs.set_option("screen_single", 0)
    hs.set_option("trace", 1)
    xs, its, es = hs.solve(y, 1e-9, budget)
    trs = hs.trace()
print("full: iter", itf, "sub:", its)
for t in range(min(len(trf["idx"]), len(trs["idx"]))):
    a = (int(trf["idx"][t]), int(trf["added"][t]), float(trf["gamma"][t]), float(trf["c_inf"][t]))
    b = (int(sub[trs["idx"][t]]), int(trs["added"][t]), float(trs["gamma"][t]), float(trs["c_inf"][t]))
    same = a[0] == b[0] and a[1] == b[1] and abs(a[2] - b[2]) <= 1e-9 * max(1.0, abs(a[2]))
    if t < 6 or not same:
        print(t, "full", a, "| sub", b, "" if same else "  <-- differs")
    if not same:
        print("column of the full pick in the sub-dictionary:", a[0] in set(sub.tolist()), " sub[0] =", int(sub[0]))
        break
