#!/bin/bash
out=gpurun_out/r3s; mkdir -p $out
WM_GEMM_LATE_DMA=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_big or conv1d" > $out/late_tests.log 2>&1; tail -3 $out/late_tests.log
for r in 0 1 0 1; do echo "WM_GEMM_LATE_DMA=$r"; WM_GEMM_LATE_DMA=$r timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep TFLOP; done > $out/bench_gemm_late.log 2>&1; cat $out/bench_gemm_late.log
for r in 0 1; do echo "WM_GEMM_LATE_DMA=$r B=32"; WM_GEMM_LATE_DMA=$r timeout 300 python scripts/bench_gemm.py 32 2>&1 | grep TFLOP; done >> $out/bench_gemm_late.log 2>&1; tail -12 $out/bench_gemm_late.log
