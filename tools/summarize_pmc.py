"""Mean of each PMC counter per launch, by kernel-name substring: tools/summarize_pmc.py <dir> <substr> [<substr> ...]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
keys = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        for k in keys:
            if k in name:
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
for k in keys:
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print("  %-28s mean %.4g over %d launches" % (c, s / max(n, 1), n))
