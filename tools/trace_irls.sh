# kernel times of IRLS construction + solves under rocprofv3 (GPU box): bash tools/trace_irls.sh
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/irt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/irt -o t -- "$PY" tools/probe_irls.py > gpurun_out/irt.log 2>&1 || exit 1
tail -4 gpurun_out/irt.log
"$PY" - <<PY
import csv,glob
f=[x for x in glob.glob("gpurun_out/irt/**/*.csv",recursive=True) if "kernel_stats" in x]
rows=sorted(csv.DictReader(open(f[0])), key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("   %-64s calls %6s avg %10.0f ns total %8.3f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"])/1e6))
PY
rm -rf gpurun_out/irt
