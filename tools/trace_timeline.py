#!/usr/bin/env python3
"""Timeline of the last solve in a rocprofv3 kernel trace: kernel, start offset, duration, gap before."""
import csv, glob, sys
path = sys.argv[1]
f = glob.glob(path + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sship" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into solves at k_sweep<float, 1 (the c0 sweep)
starts = [i for i, r in enumerate(rows) if "k_sweep<float, 1" in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
tot = {}
gaps = 0.0
for r in rows[i0:i1]:
    name = r["Kernel_Name"].split("(")[0].replace("void sship::", "").split("<")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    gap = (s - prev_end) / 1e3
    if d > 3.0 or gap > 3.0:
        print("%9.1f us  %-16s dur %8.1f  gap %6.1f" % ((s - t0) / 1e3, name, d, gap))
    tot[name] = tot.get(name, 0.0) + d
    if gap > 0:
        gaps += gap
    prev_end = max(prev_end, e)
print("solve span %.1f us; kernel time by name:" % ((prev_end - t0) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("   %-16s %9.1f us" % (k, v))
print("   gaps            %9.1f us" % gaps)
