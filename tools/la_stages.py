#!/usr/bin/env python3
"""Decodes the SS_HIP_LA_DEBUG dump: per-iteration stage times of k_la_iter's last workgroup."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)[:1024]
rows = [r for r in range(1, 1024) if a[r, 0] != 0 and a[r, 6] > a[r, 0]]
names = ["phase1 (c,q)", "barrier", "phase2 (scan)", "ticket", "select", "update"]
d = np.array([[(int(a[r, k + 1]) - int(a[r, k])) / 100.0 for k in range(6)] for r in rows])   # 100 MHz -> us
print("iterations recorded:", len(rows))
for k, nme in enumerate(names):
    print("  %-14s mean %6.2f us   min %6.2f   max %6.2f" % (nme, d[:, k].mean(), d[:, k].min(), d[:, k].max()))
print("  total          mean %6.2f us" % d.sum(1).mean())
if len(sys.argv) > 2:
    for r, row in zip(rows, d):
        print(r, " ".join("%6.2f" % v for v in row), "blk", int(a[r, 7]))
