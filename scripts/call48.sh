#!/bin/bash
# key-range split count of the decode cross-attention: by batch size (default) against two classes only (8 below 512 (b, h) pairs, else 1)
out=gpurun_out/r3at; mkdir -p $out
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for b in 4 8 12 16 24 32 48; do for pol in 0 -1; do
  name=b${b}_pol${pol}
  WM_CROSS_NSPLIT=$pol timeout 600 python bench.py $common --batch $b > $out/bench_$name.json 2> $out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$name", d["value"], "tok/s; decode step", r.get("decode_step_ms"), "loop", r.get("decode_loop_ms"))
except Exception as e: print("$name failed", e)
PY
done; done
