#!/bin/bash
# the prefetched encoder in chunks that take the whole chip once the decode loop has ended: chunk sizes x CU budgets
out=gpurun_out/r3ar; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -k "prefetch" > $out/prefetch_tests.log 2>&1; tail -3 $out/prefetch_tests.log
common="--steps 4 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced"
for cfg in "c0_e96|--encoder-cus 96|WM_PREFETCH_CHUNK=0" "c24_e96|--encoder-cus 96|WM_PREFETCH_CHUNK=24" "c24_e64|--encoder-cus 64|WM_PREFETCH_CHUNK=24" "c24_e128|--encoder-cus 128|WM_PREFETCH_CHUNK=24" "c48_e96|--encoder-cus 96|WM_PREFETCH_CHUNK=48" "c12_e96|--encoder-cus 96|WM_PREFETCH_CHUNK=12" "c24_e80|--encoder-cus 80|WM_PREFETCH_CHUNK=24" "c0_e96b|--encoder-cus 96|WM_PREFETCH_CHUNK=0" "seq|--encoder-cus 0|"; do
  IFS='|' read name args envs <<< "$cfg"
  env $envs timeout 600 python bench.py $common $args > $out/bench_$name.json 2> $out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$name", d["value"], "tok/s; ms/step", d["ms_per_step"], "loop alone", r.get("decode_loop_ms"), "beside", r.get("decode_loop_beside_encoder_ms"), "enc", (r.get("encoder") or {}).get("ms"))
except Exception as e: print("$name failed", e)
PY
done
