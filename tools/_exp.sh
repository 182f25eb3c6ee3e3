for k in 60 90 128; do echo "== k $k"; bash tools/trace_screen_kernels.sh tools/probe_res64.py --solves 4 --k $k 2>&1 | grep "configs\|k_scr_gemm<4>\|k_scr_first\|k_res_solve<double, false"; done
