#!/bin/bash
# four utterance groups once more, now that the graphs are replayed node by node
out=gpurun_out/r3aq; mkdir -p $out
common="--warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for cfg in "b32_g4|--steps 2 --batch 32 --groups 4" "b64_g4|--steps 2 --batch 64 --groups 4" "b576_g4|--steps 2 --groups 4" "b576_g3|--steps 2 --groups 3"; do
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
