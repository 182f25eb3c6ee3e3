#!/usr/bin/env python3
"""The screened form at configs[1] size (A 8192 x 65536 fp32) against the default engine on the same signals: k from 4 to 72,
signed coefficients, noise, both modes.  Certified signals must have the default engine's iteration count and support and its
coefficients to rounding; handed-back ones are that engine's bit for bit by construction (checked).

    python tools/stress_screen_full.py [signals]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-solvers_amd", "python"))
import sship  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
M, N = 8192, 65536
dev = torch.device("cuda", 0)
A_host = np.random.default_rng(1234).standard_normal((M, N), dtype=np.float32)
A_host /= np.float32(np.sqrt(M))
A = torch.from_numpy(A_host).to(dev)
del A_host
rng = np.random.default_rng(77)
bad = cert = redone = 0
x1 = torch.zeros(N, device=dev)
x0 = torch.zeros(N, device=dev)
worst_head = 0.0
for fixes in (0, 1):
    with sship.Homotopy(A, device=0) as h:
        if fixes:
            h.set_option("tie_guard", 1)
            h.set_option("zero_on_removal", 1)
        for s in range(count // 2):
            k = int(rng.integers(4, 73))
            signed = bool(rng.integers(0, 4) == 0)
            noise = float(rng.choice([0.0, 0.0, 0.0, 1e-4, 1e-3, 1e-2]))
            sup = np.sort(rng.choice(N, k, replace=False))
            coef = (1.0 + np.abs(rng.standard_normal(k))) * (rng.choice([-1.0, 1.0], k) if signed else 1.0)
            y = (A[:, torch.from_numpy(sup).to(dev)].double() @ torch.from_numpy(coef).to(dev))
            if noise:
                y = y + noise * torch.from_numpy(rng.standard_normal(M)).to(dev)
            y = y.float().contiguous()
            budget = 3 * k + 8
            h.set_option("screen_single", 1)
            h.reset_stats()
            _, it1, e1 = h.solve(y, 1e-3, budget, out=x1)
            st = h.stats()
            h.set_option("screen_single", 0)
            _, it0, e0 = h.solve(y, 1e-3, budget, out=x0)
            c = st["screen_signals"] == 1
            cert += c
            redone += st["screen_redone"]
            scale = float(x0.abs().max().item())
            d = float((x1 - x0).abs().max().item()) / max(scale, 1e-30)
            same_sup = bool(((x1 != 0) == (x0 != 0)).all().item())
            if c:
                ok = it1 == it0 and same_sup and d <= 2e-5
                worst_head = max(worst_head, st["screen_headroom"])
            else:
                ok = it1 == it0 and bool(torch.equal(x1, x0))
            if not ok:
                bad += 1
                print("BAD fixes %d k %d signed %d noise %g certified %d: iter %d / %d, same support %s, max diff %.2e, headroom %.3f" % (
                    fixes, k, signed, noise, c, it1, it0, same_sup, d, st["screen_headroom"]), flush=True)
print("signals %d: certified %d, handed back %d, bad %d; largest headroom among the certified %.3f" % (count // 2 * 2, cert, redone, bad, worst_head))
sys.exit(1 if bad else 0)
