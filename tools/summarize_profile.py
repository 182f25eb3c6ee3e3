#!/usr/bin/env python3
"""Summarises a gpurun_out/prof_<tag>/ directory produced by tools/profile_bench.sh into
profiles/<tag>_* (tracked): per-kernel durations from the rocprofv3 kernel trace and the
HBM traffic of the fused sweep from the PMC passes.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE is reported in KiB and counts
exactly half of the bytes of a wide coalesced streaming read, so
    read bytes = FETCH_SIZE * 1024 * 2;     write bytes = WRITE_SIZE * 1024.
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


KERNEL_KEY = "k_gemm32_tn_f32<128, 256, 3"     # dominant kernel whose traffic is priced (argv[2] overrides): the early form's 32-column pass


def main():
    global KERNEL_KEY
    tag = sys.argv[1]
    if len(sys.argv) > 2:
        KERNEL_KEY = sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    out = {"tag": tag}
    lines = ["# rocprofv3 summary `%s`" % tag, "",
             "command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras`", ""]

    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if trace:
        rows = list(csv.DictReader(open(trace[0])))
        per = {}
        for r in rows:
            n = short(r["Kernel_Name"])
            if "sship" not in n and "k_relayout" not in n:
                continue
            per.setdefault(n, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        lines += ["| kernel | launches | mean us (all) | launches doing work | mean us (working) |", "|---|---|---|---|---|"]
        kern = {}
        for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            # launches enqueued after the device raised `done` return at once (a few us)
            work = [x for x in v if x > 20000] if ("k_sweep" in n or "k_gemm32" in n) else v
            kern[n] = {"launches": len(v), "mean_us_all": sum(v) / len(v) / 1e3,
                       "working": len(work), "mean_us_working": sum(work) / max(1, len(work)) / 1e3}
            lines.append("| `%s` | %d | %.2f | %d | %.2f |" % (n, len(v), kern[n]["mean_us_all"],
                                                            len(work), kern[n]["mean_us_working"]))
        out["kernels"] = kern
        lines.append("")
        # timeline of one timed solve (kernel trace timestamps, us from the start of its A^T y sweep): the second
        # hardware queue carries the passes that run beside the speculative launch
        # (a solve starts with its first pass over A: k_scr_first — the screened form with both passes over the fp16 copy — or k_sweep)
        first_pass = "k_scr_first" if any("k_scr_first" in r["Kernel_Name"] for r in rows) else "k_sweep"
        starts = [i for i, r in enumerate(rows) if first_pass in r["Kernel_Name"]]
        if len(starts) >= 4:
            a = starts[len(starts) // 2]
            t0 = int(rows[a]["Start_Timestamp"])
            lines += ["## Timeline of one solve (us from the start of `%s`; queue = hardware queue id)" % first_pass, "",
                      "| start | end | queue | kernel |", "|---|---|---|---|"]
            i = a
            tl = []
            while i < len(rows) and (i == a or ("k_sweep" not in rows[i]["Kernel_Name"] and "k_scr_first" not in rows[i]["Kernel_Name"])):
                r = rows[i]
                tl.append(((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"],
                           short(r["Kernel_Name"]).replace("sship::", "")))
                i += 1
            for t in sorted(tl):
                lines.append("| %.1f | %.1f | %s | `%s` |" % t)
            solo = [t for t in tl if "k_la_persist<true>" in t[3]]
            pas = [t for t in tl if "k_gemm32" in t[3] and solo and t[2] != solo[0][2]]
            if solo and pas:
                ov = sum(max(0.0, min(solo[0][1], q[1]) - max(solo[0][0], q[0])) for q in pas)
                lines += ["", "The speculative launch `k_la_persist<true>` runs %.0f..%.0f us; the 32-column passes on the other "
                          "queue overlap it for %.0f us of their %.0f us." % (solo[0][0], solo[0][1], ov, sum(q[1] - q[0] for q in pas))]
                out["overlap_us"] = ov
            lines.append("")

    # kernels whose HBM traffic is priced: (key in traffic.json, substring of the kernel name, algorithmic bytes per launch or None = from the bench line)
    if len(sys.argv) > 2:
        targets = [("gemm32" if "gemm32" in KERNEL_KEY else "sweep2", KERNEL_KEY, None)]
    else:
        # the screened form of a single signal (csrc/screen.hip): its two passes over the fp16 copy of A, the fp32 GEMV c = A^T y (untimed run)
        # (the ranking pass: over the fp8 copy, k_scr_first8 — or over the fp16 copy, k_scr_first, where option screen_first8 is off;
        # whichever did not run has no rows in the counter files and is skipped)
        targets = [("first16", "k_scr_first8<", 8192 * 65536 * 1 + 8192 * 4 + 65536 * 4),
                   ("first16", "k_scr_first<", 8192 * 65536 * 2 + 8192 * 4 + 65536 * 4),
                   ("sweep1", "k_sweep<float, 1", 8192 * 65536 * 4 + 8192 * 4 + 65536 * 4),
                   ("screen", "k_scr_gemm", 8192 * 65536 * 2 + 96 * 8192 * 2 + 65536 * 4),
                   ("gemm32", "k_gemm32_tn_f32<128, 256, 3", 57344 * 8192 * 4 + 32 * 8192 * 4 + 32 * 57344 * 4)]
    for key, kkey, alg_fixed in targets:
        traffic = {}
        for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            f = glob.glob(os.path.join(src, "pmc_" + kind, "*", "*_counter_collection.csv"))
            if not f:
                continue
            rows = list(csv.DictReader(open(f[0])))
            vals = [float(r["Counter_Value"]) for r in rows
                    if kkey in r["Kernel_Name"] and r["Counter_Name"] == counter]
            # drop the no-op launches (solve already finished): they move (almost) nothing
            thresh = 0.5 * max(vals) if vals else 0
            vals = [v for v in vals if v >= thresh]
            if vals:
                traffic[counter] = {"launches": len(vals), "mean_raw_KiB": statistics.mean(vals)}
        if not traffic:
            continue
        rd = traffic.get("FETCH_SIZE", {}).get("mean_raw_KiB", 0.0) * 1024 * 2
        wr = traffic.get("WRITE_SIZE", {}).get("mean_raw_KiB", 0.0) * 1024
        out[key + "_hbm_read_bytes_per_launch"] = rd
        out[key + "_hbm_write_bytes_per_launch"] = wr
        out[key + "_hbm_bytes_per_launch"] = rd + wr
        out.setdefault("pmc_raw", {})[key] = traffic
        alg = alg_fixed
        if alg is None:
            nrhs = 32 if key == "gemm32" else 2
            alg = 8192 * 65536 * 4 + nrhs * 8192 * 4 + nrhs * 65536 * 4
            # (the bench line of the traced run knows the timed launch's own algorithmic bytes — the main launch of a pass that
            # is dealt out by shader engine covers 57344 of the 65536 columns)
            blog = os.path.join(src, "bench_trace.log")
            if os.path.exists(blog):
                for ln in open(blog):
                    if ln.startswith("{"):
                        try:
                            alg = int(json.loads(ln)["roofline"]["bytes_per_launch"])
                        except Exception:
                            pass
        lines += ["## HBM traffic of `%s` (PMC, separate passes)" % kkey, "",
                  "- FETCH_SIZE mean %.1f KiB x 1024 x 2 (gfx950 correction) = %.0f B read (%d launches)" % (
                      traffic.get("FETCH_SIZE", {}).get("mean_raw_KiB", 0.0), rd, traffic.get("FETCH_SIZE", {}).get("launches", 0)),
                  "- WRITE_SIZE mean %.1f KiB x 1024 = %.0f B written" % (
                      traffic.get("WRITE_SIZE", {}).get("mean_raw_KiB", 0.0), wr),
                  "- algorithmic bytes per launch: %d; traffic / algorithmic = %.4f" % (alg, (rd + wr) / alg), ""]
        tpath = os.path.join(dst, "traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        tj[key + "_hbm_bytes_per_launch"] = rd + wr
        tj[key + "_source"] = "profiles/%s_summary.md" % tag
        json.dump(tj, open(tpath, "w"), indent=1)
    log = os.path.join(src, "bench_trace.log")
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("{"):
                lines += ["## bench line of the traced run", "", "```", ln.strip(), "```", ""]
    open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(lines))
    json.dump(out, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
