# kernel times of a configs[1] solve in the screened form under rocprofv3 (tools/probe_screen.py --no-small; PROBE_MODES picks the
# (screen_single, screen_first16) rounds, e.g. PROBE_MODES="1,1;1,1")      usage (GPU box): bash tools/trace_screen_kernels.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/f16v
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f16v -o t -- python tools/probe_screen.py --no-small > gpurun_out/f16v.log 2>&1 || exit 1
grep "configs\[1\]" gpurun_out/f16v.log
python - <<PY
import csv,glob
f=[x for x in glob.glob("gpurun_out/f16v/**/*.csv",recursive=True) if "kernel_stats" in x]
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in ("k_scr_", "k_sgram", "k_sub_s", "k_sweep<")):
        print("   ", r["Name"][:70], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/f16v
