#!/bin/bash
out=gpurun_out/r3l; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log
bash scripts/ab_bench.sh r3l "default|WM_X=0|" "mt1|WM_ROWS_MT=1|" "nw4|WM_ROWS_NW=4|" "mt1nw4|WM_ROWS_MT=1 WM_ROWS_NW=4|"
