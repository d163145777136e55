#!/bin/bash
out=gpurun_out/r3m; mkdir -p $out
timeout 600 python scripts/chain_probe.py --beside-encoder 96 > $out/chain_beside96.log 2>&1; tail -13 $out/chain_beside96.log
timeout 600 python scripts/chain_probe.py --beside-encoder 48 > $out/chain_beside48.log 2>&1; tail -13 $out/chain_beside48.log
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --encoder-cus 0"
for v in "32 1 8" "32 2 8" "32 4 8" "32 2 16" "32 1 32" "64 2 8" "64 3 8" "64 4 8" "64 4 16" "8 1 8" "1 1 8"; do set -- $v
  WM_SMALL_PATH=$3 timeout 600 python bench.py $common --batch $1 --groups $2 > $out/b$1_g$2_s$3.json 2> $out/b$1_g$2_s$3.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/b$1_g$2_s$3.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("B=$1 groups=$2 small_path_rows=$3:", d["value"], "tok/s whole job; decode step", r.get("decode_step_ms"), "ms; loop", r.get("decode_loop_ms"))
except Exception as e: print("B=$1 groups=$2 small=$3 failed", e)
PY
done
