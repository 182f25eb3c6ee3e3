# screening pass (k_scr_gemm) and first pass (k_scr_first) times per row-offset multiplier SS_HIP_SCR_SKEW (GPU box): bash tools/tune_scr_skew.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${SKEWS:-29 13 21 37 45 53}; do
  export SS_HIP_SCR_SKEW=$v PROBE_MODES="1,1"
  echo "skew $v: $(timeout -k 10 120 python tools/probe_screen.py --no-small 2>&1 | grep -E "screening pass|configs\[1\]" | tr '\n' ' ' | cut -c1-260)"
done
