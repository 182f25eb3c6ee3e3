#!/usr/bin/env python3
"""GPU probe: run one speculative solve with SS_HIP_SOLO_DEBUG and print which logged breakpoints failed."""
import os, sys, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
path = "/tmp/solo_dbg.bin"
os.environ["SS_HIP_SOLO_DEBUG"] = path
import sship
m, n, k = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (96, 700, 8)
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5000 + m
rng = np.random.default_rng(seed)
A = (rng.standard_normal((m, n)) / np.sqrt(m)).astype(np.float32)
x0 = np.zeros(n); sup = np.sort(rng.choice(n, k, replace=False)); x0[sup] = 1.0 + np.abs(rng.standard_normal(k))
y = (A.astype(np.float64) @ x0).astype(np.float32)
with sship.Homotopy(A) as h:
    h.set_option("la_fused", 3)
    if os.environ.get("SOLO_DEBUG_FULLG"):
        h.set_option("gram_full_after", 1)
        h.set_option("solo_full_gram", 1)
        h.solve(y, 1e-3, 2 * k + 8)
    h.reset_stats()
    x, it, e = h.solve(y, 1e-3, 2 * k + 8)
    print("iter", it, "stats", {k_: v for k_, v in h.stats().items() if "solo" in k_})
if not os.path.exists(path):
    print("no failure dump"); sys.exit(0)
raw = open(path, "rb").read()
nlog, nvwg, ew, hw = struct.unpack("4I", raw[:16])
off = 16
lw = hw + 128 * ew
lg = np.frombuffer(raw, np.uint32, lw, off); off += 4 * lw
vm = np.frombuffer(raw, np.uint32, 128 * nvwg, off).reshape(128, nvwg); off += 4 * 128 * nvwg
vn = np.frombuffer(raw, np.uint64, 128 * nvwg, off).reshape(128, nvwg)
print("nlog", nlog, "nvwg", nvwg)
sub = lg[:256]; rows = lg[256:512]
print("subset size", (sub != 0xffffffff).sum(), "first", sub[:12], "rows", rows[:12].astype(np.int32))
f32 = lambda u: np.array([u], np.uint32).view(np.float32)[0]
for kk in range(nlog):
    e = lg[hw + kk * ew: hw + (kk + 1) * ew]
    K, flags, rnd, idx = e[0], e[1], e[2], e[3]
    mx = vm[kk].max(); mn = vn[kk].min()
    ok_l = (e[4] & 0x7fffffff) == mx
    tg, ti = f32(np.uint32(mn >> np.uint64(32))), int(mn & np.uint64(0xffffffff))
    print("entry %2d K=%2d flags=%d round=%d lam log=%.7g true=%.7g %s | pick log=(%.7g,%d) supp=(%.7g,%d) true_off=(%.7g,%d)" % (
        kk, K, flags, rnd, f32(e[4]), f32(mx), "ok" if ok_l else "LAMBDA-MISMATCH", f32(e[5]), idx, f32(e[6]), int(e[7]), tg, ti))
    pos = e[8 + 96: 8 + 96 + K]
    print("     gam", e[8:8 + K], "pos", pos)
