#!/bin/bash
out=gpurun_out/r3k; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_rows.py -x -q > $out/rows_tests.log 2>&1; tail -3 $out/rows_tests.log
timeout 300 python scripts/bench_rows.py 64 192 > $out/bench_rows.log 2>&1; cat $out/bench_rows.log
timeout 600 python scripts/chain_probe.py --rows-path 1 > $out/chain_rows.log 2>&1; tail -13 $out/chain_rows.log
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --encoder-cus 0"
for b in 12 16 32 64 128; do for r in 0 1; do
  WM_ROWS_PATH=$r timeout 600 python bench.py $common --batch $b > $out/b${b}_rows$r.json 2> $out/b${b}_rows$r.err
  python - <<PY
import json
d=json.loads(open("$out/b${b}_rows$r.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("B=$b rows_path=$r", d["value"], "tok/s; decode step", r.get("decode_step_ms"), "ms; loop", r.get("decode_loop_ms"))
PY
done; done
bash scripts/ab_bench.sh r3k "rows|WM_ROWS_PATH=1|"
