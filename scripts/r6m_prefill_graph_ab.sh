#!/bin/bash
# round 6: the language pass + prefill replayed from graphs (WhisperDecoding.graph_prefill) against eager issue, interleaved
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round6.py -q -x 2>&1 | tail -3
for round in 1 2; do
for b in 1 4 8 16 32 64 256; do
  for gp in 1 0; do
    line=$(WM_GRAPH_PREFILL=$gp python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-measure-traffic --no-roofline --length-dist forced 2>/dev/null | grep '^{' | tail -1)
    python - "$b" "$gp" "$round" "$line" <<'PY'
import json, sys
b, gp, rd, line = sys.argv[1:5]
d = json.loads(line); p = d["pipeline"]
print(f"round {rd} batch {b} graph_prefill {gp}: {d['value']} tokens/s, {d['ms_per_step']} ms per step; cross-K/V + language {p['cross_kv_and_language_ms']} ms, first token after encoder {p['first_token_after_encoder_ms']} ms, loop {p['prefill_and_decode_loop_ms']} ms")
PY
  done
done
done
