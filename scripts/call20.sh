#!/bin/bash
out=gpurun_out/r3x; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or conv" > $out/kernel_tests.log 2>&1; tail -3 $out/kernel_tests.log
timeout 300 python scripts/bench_gemm.py 128 2>&1 | grep TFLOP > $out/bench_gemm.log; cat $out/bench_gemm.log
REPS=4 timeout 300 python scripts/bench_gemm.py 576 2>&1 | grep TFLOP >> $out/bench_gemm.log; tail -5 $out/bench_gemm.log
timeout 600 python scripts/stage_times.py --batch 576 --decode-steps 8 --reps 2 2>&1 | grep rep
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
