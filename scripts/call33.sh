#!/bin/bash
# third session of round 3: greedy kernel scan, four-wave self-attention, merge inside the output projection
out=gpurun_out/r3af; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py -x -q -k "greedy or attn_decode_self or merge_in_projection or wave_forms or live_rows or ragged" > $out/new_tests.log 2>&1; tail -5 $out/new_tests.log
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced"
for cfg in "b1|--batch 1|" "b1_old|--batch 1|WM_SELF_WAVES=1 WM_MERGE_IN_PROJ=0" "b1_w4only|--batch 1|WM_MERGE_IN_PROJ=0" "b8|--batch 8|" "b8_old|--batch 8|WM_SELF_WAVES=1" "b32|--batch 32|" "b32_w4|--batch 32|WM_SELF_WAVES=4" "b576_w1|--steps 3|WM_SELF_WAVES=1" "b576_w4|--steps 3|WM_SELF_WAVES=4"; do
  IFS='|' read name args envs <<< "$cfg"
  env $envs timeout 600 python bench.py $common $args > $out/bench_$name.json 2> $out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$name", d["value"], "tok/s; ms/step", d["ms_per_step"], "decode step", r.get("decode_step_ms"), "loop", r.get("decode_loop_ms"))
except Exception as e: print("$name failed", e)
PY
done
