#!/bin/bash
out=gpurun_out/r3ac; mkdir -p $out
for c in fp16 int8wo int8kv int8 int4 int8x; do
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --config $c > $out/bench_line_config_$c.json 2> $out/bench_line_config_$c.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_line_config_$c.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$c", d["value"], "tok/s;", "K/V launch", r.get("achieved"), "GB/s alone; step", r.get("decode_step_ms"), "ms; weights", d["hbm_bytes_resident"]["engine_weights"], "in use", d["hbm_bytes_resident"]["device_in_use"])
except Exception as e: print("$c failed", e)
PY
done
