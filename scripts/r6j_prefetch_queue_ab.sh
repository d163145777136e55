#!/bin/bash
# round 6: the prefetched encoder pass on a pooled torch stream against a hardware queue of its own (WM_PREFETCH_QUEUE), interleaved
export TMPDIR=/tmp
for round in 1 2; do
for b in 1 2 8 32 576; do
  for q in pooled dedicated; do
    st=3; [ $b = 576 ] && st=4
    line=$(WM_PREFETCH_QUEUE=$q python bench.py --batch $b --steps $st --warmup 1 --no-cpu-baseline --no-measure-traffic --no-roofline 2>/dev/null | grep '^{' | tail -1)
    python - "$b" "$q" "$round" "$line" <<'PY'
import json, sys
b, q, rd, line = sys.argv[1:5]
d = json.loads(line); p = d["pipeline"]; s = d["second_figure"] or {}
print(f"round {rd} batch {b} {q}: {d['value']} tokens/s, {d['ms_per_step']} ms per step; other {p['other_ms']} ms, first token after encoder {p['first_token_after_encoder_ms']} ms, loop {p['prefill_and_decode_loop_ms']} ms, collect wait {p['collect_wait_ms']}; second figure {s.get('ms_per_batch')} / pipelined {s.get('ms_per_batch_pipelined')} ms")
PY
  done
done
done
