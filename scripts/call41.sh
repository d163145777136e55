#!/bin/bash
# where the small-batch switch sits after the third session's changes: B = 12 / 16 / 24 / 32, groups x WM_SMALL_PATH
out=gpurun_out/r3an; mkdir -p $out
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for cfg in "b12_g1_s8|--batch 12 --groups 1|WM_SMALL_PATH=8" "b12_g1_s16|--batch 12 --groups 1|WM_SMALL_PATH=16" "b12_g2_s8|--batch 12 --groups 2|WM_SMALL_PATH=8" \
           "b16_g1_s8|--batch 16 --groups 1|WM_SMALL_PATH=8" "b16_g1_s16|--batch 16 --groups 1|WM_SMALL_PATH=16" "b16_g2_s8|--batch 16 --groups 2|WM_SMALL_PATH=8" \
           "b24_g2_s8|--batch 24 --groups 2|WM_SMALL_PATH=8" "b24_g2_s16|--batch 24 --groups 2|WM_SMALL_PATH=16" "b24_g3_s8|--batch 24 --groups 3|WM_SMALL_PATH=8" \
           "b32_g2_s8|--batch 32 --groups 2|WM_SMALL_PATH=8" "b32_g2_s16|--batch 32 --groups 2|WM_SMALL_PATH=16" "b32_g1_s8|--batch 32 --groups 1|WM_SMALL_PATH=8" "b32_g3_s16|--batch 32 --groups 3|WM_SMALL_PATH=16"; do
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
