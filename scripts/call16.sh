#!/bin/bash
out=gpurun_out/r3u; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q > $out/kernel_tests.log 2>&1; tail -3 $out/kernel_tests.log
for r in 0 1; do echo "WM_GEMM_ROUND1=$r"; WM_GEMM_ROUND1=$r timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep TFLOP; done > $out/bench_gemm_epilogue.log 2>&1; cat $out/bench_gemm_epilogue.log
timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep TFLOP >> $out/bench_gemm_epilogue.log; tail -5 $out/bench_gemm_epilogue.log
timeout 600 python scripts/stage_times.py --batch 576 --decode-steps 8 --reps 2 2>&1 | grep rep
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
