#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + PMC passes of the bench command.
# usage: tools/profile_bench.sh <tag>     (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (the interpreter binary itself after `--`: no PATH shim / wrapper hop behind the profiler's preloaded library)
PY=$(python3 -c 'import sys,os;print(os.path.realpath(sys.executable))')
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$PY" "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_trace.log" 2>&1 || echo "trace run failed rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- "$PY" "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --batch 0 > "$OUT/bench_pmc_fetch.log" 2>&1 || echo "pmc fetch run failed rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- "$PY" "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --batch 0 > "$OUT/bench_pmc_write.log" 2>&1 || echo "pmc write run failed rc=$?"
find "$OUT" -name "*.csv" | head -50
