#!/bin/bash
# round-3 GPU call 2: new kernel tests, micro-benchmarks, whole suite, A/B of the decode paths
out=gpurun_out/r3h; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rows.py -x -q > $out/rows_tests.log 2>&1; tail -15 $out/rows_tests.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q > $out/kernel_tests.log 2>&1; tail -3 $out/kernel_tests.log
timeout 300 python scripts/bench_rows.py 16 32 64 192 > $out/bench_rows.log 2>&1; cat $out/bench_rows.log
timeout 300 python scripts/bench_gemm.py 128 > $out/bench_gemm_128.log 2>&1; cat $out/bench_gemm_128.log
timeout 300 python scripts/bench_attn_encoder.py > $out/bench_attn.log 2>&1; cat $out/bench_attn.log
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log
bash scripts/ab_bench.sh r3h "splitk|WM_ROWS_PATH=0|" "rows|WM_ROWS_PATH=1|" "rows_seq|WM_ROWS_PATH=1|--encoder-cus 0" "splitk_seq|WM_ROWS_PATH=0|--encoder-cus 0"
