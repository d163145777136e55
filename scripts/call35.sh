#!/bin/bash
# DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (graph nodes enqueued one by one instead of as pre-built AQL packets) at every batch size
out=gpurun_out/r3ah; mkdir -p $out
common="--warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced"
for cfg in "b8_cap1|--steps 2 --encoder-cus 0 --batch 8|" "b8_cap0|--steps 2 --encoder-cus 0 --batch 8|DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "b32_cap1|--steps 2 --encoder-cus 0 --batch 32|" "b32_cap0|--steps 2 --encoder-cus 0 --batch 32|DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "b576_cap1|--steps 3|" "b576_cap0|--steps 3|DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "b576seq_cap1|--steps 3 --encoder-cus 0|" "b576seq_cap0|--steps 3 --encoder-cus 0|DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "b1_cap0|--steps 2 --encoder-cus 0 --batch 1|DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "b1_cap1|--steps 2 --encoder-cus 0 --batch 1|"; do
  IFS='|' read name args envs <<< "$cfg"
  env $envs timeout 600 python bench.py $common $args > $out/bench_$name.json 2> $out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$name", d["value"], "tok/s; ms/step", d["ms_per_step"], "decode step", r.get("decode_step_ms"), "loop", r.get("decode_loop_ms"), "beside", r.get("decode_step_beside_encoder_ms"))
except Exception as e: print("$name failed", e)
PY
done
