#!/bin/bash
out=gpurun_out/r3z; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn_decode_self or quantize" > $out/kernel_tests.log 2>&1; tail -3 $out/kernel_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
common="--steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --encoder-cus 0"
for b in 1 8 32; do timeout 600 python bench.py $common --batch $b > $out/bench_b$b.json 2> $out/bench_b$b.err; python scripts/sumline.py $out/bench_b$b.json; done
timeout 600 python scripts/chain_probe.py --rows-path 1 > $out/chain_rows.log 2>&1; tail -13 $out/chain_rows.log
bash scripts/ab_bench.sh r3z "final|WM_X=0|"
