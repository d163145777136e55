#!/bin/bash
# key-range split count of the decode cross-attention, forced, at small batches (default: ceil(512 / (rows x heads)), at most 8)
out=gpurun_out/r3au; mkdir -p $out
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for cfg in "b8|1 2 3 4 6" "b16|1 2 3 4" "b32|1 2 3" "b64|1 2" "b2|2 4 8" "b1|4 8"; do
  IFS='|' read bname splits <<< "$cfg"; b=${bname#b}
  for n in $splits; do
    name=${bname}_n$n
    WM_CROSS_NSPLIT=$n timeout 600 python bench.py $common --batch $b > $out/bench_$name.json 2> $out/bench_$name.err
    python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$name", d["value"], "tok/s; decode step", r.get("decode_step_ms"), "loop", r.get("decode_loop_ms"))
except Exception as e: print("$name failed", e)
PY
  done
done
