#!/bin/bash
out=gpurun_out/r3p; mkdir -p $out
for nt in 0 1 4 5; do echo "WM_GEMM_NT=$nt"; WM_GEMM_NT=$nt timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep TFLOP; done > $out/bench_gemm_nt.log 2>&1; cat $out/bench_gemm_nt.log
bash scripts/ab_bench.sh r3p "nt0|WM_GEMM_NT=0|" "nt1|WM_GEMM_NT=1|" "nt4|WM_GEMM_NT=4|" "nt5|WM_GEMM_NT=5|"
