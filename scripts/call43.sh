#!/bin/bash
# group counts above 32 utterances: three groups of <= 16 rows (fused small-batch path) against two larger groups (split-K chain)
out=gpurun_out/r3ap; mkdir -p $out
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for cfg in "b40_g2|--batch 40 --groups 2" "b40_g3|--batch 40 --groups 3" "b48_g2|--batch 48 --groups 2" "b48_g3|--batch 48 --groups 3" "b64_g2|--batch 64 --groups 2" "b64_g3|--batch 64 --groups 3" "b96_g2|--batch 96 --groups 2" "b96_g3|--batch 96 --groups 3" "b120_g2|--batch 120 --groups 2" "b120_g3|--batch 120 --groups 3"; do
  IFS='|' read name args <<< "$cfg"
  timeout 600 python bench.py $common $args > $out/bench_$name.json 2> $out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$name", d["value"], "tok/s; decode step", r.get("decode_step_ms"), "loop", r.get("decode_loop_ms"))
except Exception as e: print("$name failed", e)
PY
done
