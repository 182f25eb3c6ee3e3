#!/usr/bin/env python3
"""Timeline view of one solve from a rocprofv3 kernel trace: per-kernel durations and the
idle gaps between consecutive kernels of the solver's stream."""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
f = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sship" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = defaultdict(list)
gap_after = defaultdict(list)
prev = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void sship::", "")
    name = name.split("<")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s < 20000 and name == "k_sweep":
        prev = None
        continue   # no-op sweep
    dur[name].append(e - s)
    if prev is not None and s - prev[1] < 200000:
        gap_after[prev[0] + " -> " + name].append(s - prev[1])
    prev = (name, e)
print("durations (us): mean / min / max / count")
for k, v in dur.items():
    print("  %-12s %8.2f %8.2f %8.2f %6d" % (k, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, len(v)))
print("gaps (us): mean / min / max / count")
tot = 0
for k, v in gap_after.items():
    print("  %-24s %8.2f %8.2f %8.2f %6d" % (k, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, len(v)))
