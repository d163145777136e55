#!/bin/bash
# where the four-wave self-attention stops paying: groups of 32 / 64 / 96 rows
out=gpurun_out/r3as; mkdir -p $out
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for cfg in "b64_w0|--batch 64|" "b64_w4|--batch 64|WM_SELF_WAVES=4" "b128_w0|--batch 128|" "b128_w4|--batch 128|WM_SELF_WAVES=4" "b192_w0|--batch 192|" "b192_w4|--batch 192|WM_SELF_WAVES=4" "b32_w0|--batch 32|" "b32_w4|--batch 32|WM_SELF_WAVES=4"; do
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
