import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sship" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = {}
seq = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void sship::", "").split("<")[0]
    seq.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
gemm = [d for n, d in seq if n == "k_gemm_tn_f32"]
print("gemm launches", len(gemm))
for i in (0, 1, 2, 3, 60, 100, 130, 131, 138, 139, 140, 141, 200, 300, 400, len(gemm) - 1):
    if i < len(gemm):
        print("  gemm #%d: %.1f us" % (i, gemm[i]))
for name in ("k_absmax", "k_scansel", "k_gramupd", "k_rp", "k_tile_skip"):
    d = [x for n, x in seq if n == name]
    if d:
        print(name, "count", len(d), "first10 mean %.1f" % (sum(d[:10]) / len(d[:10])), "mid mean %.1f" % (sum(d[100:110]) / max(1, len(d[100:110]))), "last10 mean %.1f us" % (sum(d[-10:]) / len(d[-10:])))
