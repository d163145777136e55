#!/bin/bash
out=gpurun_out/r3al; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cross" > $out/cross_tests.log 2>&1; tail -3 $out/cross_tests.log
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for cfg in "b1_pre1|--batch 1|" "b1_pre0|--batch 1|WM_CROSS_PRELOAD=0" "b4_pre1|--batch 4|" "b4_pre0|--batch 4|WM_CROSS_PRELOAD=0" "b1_pre1b|--batch 1|" "b1_pre0b|--batch 1|WM_CROSS_PRELOAD=0"; do
  IFS='|' read name args envs <<< "$cfg"
  env $envs timeout 600 python bench.py $common $args > $out/bench_$name.json 2> $out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$name", d["value"], "tok/s; decode step", r.get("decode_step_ms"), "loop", r.get("decode_loop_ms"))
except Exception as e: print("$name failed", e)
PY
done
