#!/bin/bash
out=gpurun_out/r3v; mkdir -p $out
for r in 0 90 0 90 45 180; do echo "WM_GEMM_SKEW=$r"; WM_GEMM_SKEW=$r timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep TFLOP; done > $out/bench_gemm_skew.log 2>&1; cat $out/bench_gemm_skew.log
for r in 0 90; do echo "WM_GEMM_SKEW=$r B=576"; REPS=4 WM_GEMM_SKEW=$r timeout 300 python scripts/bench_gemm.py 576 2>&1 | grep TFLOP; done >> $out/bench_gemm_skew.log 2>&1; tail -12 $out/bench_gemm_skew.log
