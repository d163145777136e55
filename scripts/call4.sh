#!/bin/bash
out=gpurun_out/r3j; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rows.py tests/test_gpu_kernels.py -x -q > $out/kernel_tests.log 2>&1; tail -5 $out/kernel_tests.log
timeout 300 python scripts/bench_rows.py 16 64 192 > $out/bench_rows.log 2>&1; cat $out/bench_rows.log
timeout 600 python scripts/chain_probe.py --rows-path 1 > $out/chain_rows.log 2>&1; tail -14 $out/chain_rows.log
timeout 600 python scripts/chain_probe.py --rows-path 0 > $out/chain_splitk.log 2>&1; tail -16 $out/chain_splitk.log
timeout 600 python scripts/chain_probe.py --rows-path 1 --batch 192 --groups 1 > $out/chain_rows_alone.log 2>&1; tail -14 $out/chain_rows_alone.log
bash scripts/ab_bench.sh r3j "rows|WM_ROWS_PATH=1|" "splitk|WM_ROWS_PATH=0|" "rows_seq|WM_ROWS_PATH=1|--encoder-cus 0"
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log
