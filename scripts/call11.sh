#!/bin/bash
# final artefacts of a code state: the driver's command, kernel statistics of the same command (sequential steps), small batches
tag=${1:-r3q}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -3 $out/gpu_tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err; python scripts/sumline.py $out/bench_line.json
d=/tmp/prof_$tag; rm -rf $d
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --encoder-cus 0 --no-cpu-baseline --no-measure-traffic --length-dist forced > $out/bench_under_profiler.json 2> $out/prof.err
find $d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/bench_kernel_stats.csv; head -14 $out/bench_kernel_stats.csv | cut -c1-150
common="--steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --encoder-cus 0"
for b in 1 8 32; do timeout 600 python bench.py $common --batch $b > $out/bench_b$b.json 2> $out/bench_b$b.err; python scripts/sumline.py $out/bench_b$b.json; done
d=/tmp/prof_b1_$tag; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $common --batch 1 > /dev/null 2> $out/prof_b1.err
find $d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/b1_kernel_stats.csv
