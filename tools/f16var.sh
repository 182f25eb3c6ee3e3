# times of the half-precision passes per variant of k_scr_first (SS_HIP_SCR_FIRST) under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-0}; do
  export SS_HIP_SCR_FIRST=$v
  rm -rf gpurun_out/f16v$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f16v$v -o t -- python tools/probe_screen.py --no-small > gpurun_out/f16v$v.log 2>&1 || exit 1
  echo "variant $v"; grep "first16 1" gpurun_out/f16v$v.log | tail -1
  python - <<PY
import csv,glob
f=[x for x in glob.glob("gpurun_out/f16v$v/**/*.csv",recursive=True) if "kernel_stats" in x]
print(f)
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in ("k_scr_", "k_sgram", "k_sub_s", "k_sweep<")):
        print("   ", r["Name"][:70], r["Calls"], r["AverageNs"])
PY
  rm -rf gpurun_out/f16v$v
done
