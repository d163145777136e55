#!/bin/bash
# the LayerNorm helper waves against round 5's gemv_small.hip (build/lab/libwm_r5gemv.so), interleaved, three rounds
export TMPDIR=/tmp
for round in 1 2 3; do
for b in 10 12 16 24 32; do
  for lib in product r5; do
    if [ $lib = r5 ]; then export WM_LIBRARY_PATH=$PWD/build/lab/libwm_r5gemv.so; else unset WM_LIBRARY_PATH; fi
    line=$(python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --encoder-cus 0 2>/dev/null | grep '^{' | tail -1)
    python - "$b" "$lib" "$round" "$line" <<'PY'
import json, sys
b, w, rd, line = sys.argv[1:5]
d = json.loads(line); r = d["roofline"]
print(f"round {rd} batch {b} gemv_small {w}: {r['decode_step_ms']} ms per token step, {d['value']} tokens/s")
PY
  done
done
done
