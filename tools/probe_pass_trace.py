"""Developer probe: where and when the workgroups of the early form's two passes run in a real configs[1] solve
(option pass_dbg_ptr: per-workgroup start / end / XCC_ID / HW_ID), beside the speculative launch and — for comparison —
with the passes first and the launch afterwards (early_probe = 1)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship
M, N, K = 8192, 65536, 64
A = np.random.default_rng(1234).standard_normal((M, N), dtype=np.float32)
A /= np.float32(np.sqrt(M))
Ad = torch.from_numpy(A).to("cuda:0")
rng = np.random.default_rng(1235)
sup = np.sort(rng.choice(N, K, replace=False))
coef = 1.0 + np.abs(rng.standard_normal(K))
y = (Ad[:, torch.from_numpy(sup).cuda()].double() @ torch.from_numpy(coef).cuda()).float().contiguous()
x = torch.zeros(N, device="cuda:0")
buf = torch.zeros(1 + 4 * 4096, dtype=torch.int64, device="cuda:0")


def report(tag):
    b = buf.cpu().numpy().astype(np.uint64)
    ne = int(b[0])
    e = b[1:1 + 4 * ne].reshape(ne, 4)
    t0, t1, where, blk = e[:, 0].astype(np.int64), e[:, 1].astype(np.int64), e[:, 2], e[:, 3]
    base = t0.min()
    order = np.argsort(t0)
    # passes = clusters of starts
    starts = (t0[order] - base) / 100.0
    cuts = [0] + [i + 1 for i in range(ne - 1) if starts[i + 1] - starts[i] > 50.0] + [ne]
    print("== %s: %d workgroup records, %d launches" % (tag, ne, len(cuts) - 1))
    for c in range(len(cuts) - 1):
        sel = order[cuts[c]:cuts[c + 1]]
        s0, s1 = (t0[sel] - base) / 100.0, (t1[sel] - base) / 100.0
        xcc = (where[sel] >> np.uint64(16)).astype(int)
        hw = (where[sel] & np.uint64(0xffff)).astype(int)
        cu = xcc * 256 + ((hw >> 13) & 7) * 32 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)
        dur = s1 - s0
        print("  launch %d: %d workgroups, starts %.1f..%.1f us, ends %.1f..%.1f us (span %.1f), duration mean %.1f max %.1f" % (
            c, len(sel), s0.min(), s0.max(), s1.min(), s1.max(), s1.max() - s0.min(), dur.mean(), dur.max()))
        counts = {}
        for v in cu:
            counts[v] = counts.get(v, 0) + 1
        hist = {}
        for v in counts.values():
            hist[v] = hist.get(v, 0) + 1
        print("    CUs used %d; workgroups per CU: %s" % (len(counts), " ".join("%dx%d" % (hist[k], k) for k in sorted(hist))))
        se = (hw >> 13) & 7
        sh = (hw >> 12) & 1
        per_se = {}
        for a, b_, c_ in zip(xcc, se, sh):
            per_se[(a, b_, c_)] = per_se.get((a, b_, c_), 0) + 1
        cus_se = {}
        for a, b_, c_, d_ in zip(xcc, se, sh, cu):
            cus_se.setdefault((a, b_, c_), set()).add(d_)
        print("    workgroups / CUs per (XCC, SE, SH): " + "  ".join("%d.%d.%d: %d/%d" % (k[0], k[1], k[2], v, len(cus_se[k])) for k, v in sorted(per_se.items())))
        for xc in range(8):
            m_ = xcc == xc
            if m_.any():
                cus = {}
                for v in cu[m_]:
                    cus[v] = cus.get(v, 0) + 1
                big = [k for k, v in cus.items() if v >= 3]
                dbig = dur[m_][np.isin(cu[m_], big)] if big else np.array([0.0])
                print("    XCC %d: %3d workgroups on %2d CUs, duration mean %.1f max %.1f, last end %.1f; CUs with 3+: %d (their workgroups: mean %.1f us)" % (
                    xc, int(m_.sum()), len(cus), dur[m_].mean(), dur[m_].max(), s1[m_].max(), len(big), dbig.mean()))


with sship.Homotopy(Ad) as h:
    del Ad
    for probe, tag in ((0, "passes beside the speculative launch"), (1, "passes first, launch afterwards")):
        h.set_option("early_probe", probe)
        for _ in range(3):
            h.solve(y, 1e-3, 256, out=x)
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        h.set_option("pass_dbg_ptr", buf.data_ptr())
        h.solve(y, 1e-3, 256, out=x)
        torch.cuda.synchronize()
        h.set_option("pass_dbg_ptr", 0)
        report(tag)
