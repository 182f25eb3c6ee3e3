#!/usr/bin/env python3
"""Decodes the SS_HIP_LA_DEBUG dump of the resident kernel: stage times of workgroup 0 per iteration."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2048, 8)
rows = [r for r in range(1, 1024) if a[r, 0] != 0 and a[r, 6] > a[r, 0] and a[r, 7] >= a[r, 6]]
names = ["q pass + stores", "lambda poll", "scan + step exchange", "pick, x, c pass, post", "u2, d, lists", "sign + direction", "c/q pass (solo)"]
d = np.array([[(int(a[r, k + 1]) - int(a[r, k])) / 100.0 for k in range(7)] for r in rows])
print("launch info (last): lds rows used %d, workgroups %d, lds rows %d, K at entry %d" % tuple(int(v) for v in a[0, :4]))
for r in range(1600, 1700):
    if a[r, 7] > a[r, 0]:
        t = [(int(a[r, q + 1]) - int(a[r, q])) / 100.0 for q in range(7)]
        print("k_la_verify with %d entries (workgroup 0, first chunk): entry %.1f  bulk copy %.1f  tables %.1f  rows %.1f  max %.1f  scan+min %.1f  rest/2nd chunk %.1f us" % tuple([r - 1600] + t))
print("iterations recorded:", len(rows))
for k, nme in enumerate(names):
    print("  %-22s mean %6.2f us   min %6.2f   max %6.2f" % (nme, d[:, k].mean(), d[:, k].min(), d[:, k].max()))
print("  total                  mean %6.2f us" % d.sum(1).mean())
gaps = [(int(a[rows[i + 1], 0]) - int(a[rows[i], 6])) / 100.0 for i in range(len(rows) - 1) if rows[i + 1] == rows[i] + 1]
if gaps:
    print("  iteration to iteration mean %6.2f us (end of one to start of the next)" % np.mean(gaps))
if len(sys.argv) > 2:
    for r, row in zip(rows, d):
        print(r, " ".join("%6.2f" % v for v in row))
