import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sship" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last solve: from the last k_la_init_pick onwards
idx = max(i for i, r in enumerate(rows) if "k_la_init_pick" in r["Kernel_Name"])
rows = rows[idx - 1:]
dur = defaultdict(list); gap = defaultdict(list); prev = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void sship::", "").replace("sship::", "").split("<")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append((e - s) / 1e3)
    if prev: gap[prev[0] + "->" + n].append((s - prev[1]) / 1e3)
    prev = (n, e)
tot = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("solve span %.1f us, kernels %d" % (tot, len(rows)))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-16s n=%4d total %8.1f us  mean %7.2f  min %6.2f max %7.2f" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
print("gaps:")
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    print("  %-34s n=%4d total %7.1f mean %5.2f" % (k, len(v), sum(v), sum(v) / len(v)))
