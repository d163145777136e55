#!/bin/bash
out=gpurun_out/r3aw; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for b in 1 2 4 6 8 12 16 24 32 64; do timeout 600 python bench.py $common --batch $b > $out/bench_b$b.json 2> $out/bench_b$b.err; python scripts/sumline.py $out/bench_b$b.json; done
