#!/usr/bin/env python3
"""Decodes the SS_HIP_LA_DEBUG dump of the resident kernel: master and worker-0 stage times."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2048, 8)
mrows = [r for r in range(1, 1024) if a[r, 0] != 0 and a[r, 6] > a[r, 0]]
names = ["publish", "wait workers", "pick+x", "slot/u1 loads", "inverse", "sign+direction"]
d = np.array([[(int(a[r, k + 1]) - int(a[r, k])) / 100.0 for k in range(6)] for r in mrows])
print("master iterations recorded:", len(mrows))
for k, nme in enumerate(names):
    print("  %-16s mean %6.2f us   min %6.2f   max %6.2f" % (nme, d[:, k].mean(), d[:, k].min(), d[:, k].max()))
print("  total            mean %6.2f us" % d.sum(1).mean())
wrows = [r for r in range(1024, 2048) if a[r, 0] != 0 and a[r, 5] > a[r, 0]]
wn = ["wait publish", "stage triples", "c,q (gram form)", "max hand-off", "scan+min offer"]
w = np.array([[(int(a[r, k + 1]) - int(a[r, k])) / 100.0 for k in range(5)] for r in wrows])
print("worker-0 ticks recorded:", len(wrows))
for k, nme in enumerate(wn):
    print("  %-16s mean %6.2f us   min %6.2f   max %6.2f" % (nme, w[:, k].mean(), w[:, k].min(), w[:, k].max()))
print("  total            mean %6.2f us" % w.sum(1).mean())
if len(sys.argv) > 2:
    for r, row in zip(mrows, d):
        print(r, " ".join("%6.2f" % v for v in row), "K", int(a[r, 7]))
