#!/bin/bash
out=gpurun_out/r3aa; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rows.py tests/test_gpu_kernels.py -x -q > $out/kernel_tests.log 2>&1; tail -3 $out/kernel_tests.log
for u in 4 2; do echo "WM_CROSS_UNR=$u"; WM_CROSS_UNR=$u timeout 300 python scripts/bench_cross_big.py 2>&1 | tail -4; done
WM_CROSS_UNR=4 timeout 600 python scripts/chain_probe.py --rows-path 1 > $out/chain_unr4.log 2>&1; tail -13 $out/chain_unr4.log
WM_CROSS_UNR=2 timeout 600 python scripts/chain_probe.py --rows-path 1 > $out/chain_unr2.log 2>&1; tail -13 $out/chain_unr2.log
bash scripts/ab_bench.sh r3aa "unr4|WM_CROSS_UNR=4|" "unr2|WM_CROSS_UNR=2|" "unr2_seq|WM_CROSS_UNR=2|--encoder-cus 0" "unr4_seq|WM_CROSS_UNR=4|--encoder-cus 0"
