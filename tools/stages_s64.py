#!/usr/bin/env python3
"""Stage timestamps of k_la_iter in the sub-context of the fp64 screened form (SS_HIP_LA_DEBUG=<file>; 100 MHz ticks)."""
import os
import subprocess
import sys

import numpy as np

path = "/tmp/la_dbg.bin"
if len(sys.argv) > 1 and sys.argv[1] == "run":
    os.environ["SS_HIP_LA_DEBUG"] = path
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.run([sys.executable, os.path.join(here, "trace_screen64.py")], check=True)
ts = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
rows = [r for r in range(1, 200) if ts[r, 0] != 0 and ts[r, 6] > ts[r, 0]]
names = ["phase 1 (c, q)", "grid barrier", "lambda + scan", "ticket", "select_toggle", "gather + inverse + direction"]
d = np.array([[float(ts[r, i + 1] - ts[r, i]) / 100.0 for i in range(6)] for r in rows])
print("rounds with stamps:", len(rows))
for i, nme in enumerate(names):
    print("%-30s mean %6.2f us   (rounds 1-32 %6.2f, 33-64 %6.2f, 65-128 %6.2f)" % (
        nme, d[:, i].mean(), d[:32, i].mean(), d[32:64, i].mean(), d[64:, i].mean()))
print("%-30s mean %6.2f us" % ("whole launch (stamped part)", d.sum(axis=1).mean()))
