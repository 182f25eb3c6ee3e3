# kernel times of a configs[1] solve in the screened form under rocprofv3 (tools/probe_screen.py --no-small; PROBE_MODES picks the
# (screen_single, screen_first16) rounds, e.g. PROBE_MODES="1,1;1,1")      usage (GPU box): bash tools/trace_screen_kernels.sh [probe script + args]
# (the interpreter binary itself goes after `--`: a PATH shim or wrapper script would be an exec hop behind the profiler's preloaded library)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/f16v
rm -rf "$OUT"
if [ $# -gt 0 ]; then PROBE=("$@"); else PROBE=(tools/probe_screen.py --no-small); fi
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- "$PY" "${PROBE[@]}" > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
grep "configs\[" "$OUT.log"
"$PY" - <<PYEOF
import csv,glob
f=[x for x in glob.glob("$OUT/**/*.csv",recursive=True) if "kernel_stats" in x]
rows=sorted(csv.DictReader(open(f[0])), key=lambda r:-float(r["TotalDurationNs"]))
for r in rows:
    if any(k in r["Name"] for k in ("k_scr_", "k_sgram", "k_sub_s", "k_sweep<", "k_res_", "k_epilogue", "k_la_reset", "k_s64")):
        print("   %-72s calls %6s avg %10.0f ns" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])))
PYEOF
rm -rf "$OUT"
