#!/bin/bash
# Runs on the GPU box: SQ counters of the fp64 lookahead pass (k_gemm32_tn_f64<32> / <64>) at configs[4] size.
# usage: tools/profile_f64_pass.sh     (outputs under gpurun_out/prof_f64/)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_f64
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d "$OUT/pmc_a" -- python3 "$REPO/tools/probe_c5_cols.py" > "$OUT/pmc_a.log" 2>&1 || echo "pmc a failed rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_b" -- python3 "$REPO/tools/probe_c5_cols.py" > "$OUT/pmc_b.log" 2>&1 || echo "pmc b failed rc=$?"
find "$OUT" -name "*.csv" | head
