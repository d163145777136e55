#!/bin/bash
# round 6: utterance groups x batch at 24-128 utterances (token step in ms; the library's default is 2 groups from 16, 3 from 128)
export TMPDIR=/tmp
for b in ${BATCHES:-24 32 48 64 96 128}; do
  for g in 0 1 2 3 4; do
    line=$(python bench.py --batch $b --groups $g --steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --encoder-cus 0 2>/dev/null | grep '^{' | tail -1)
    python - "$b" "$g" "$line" <<'PY'
import json, sys
b, g, line = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(line); r = d["roofline"]
    print(f"batch {b} groups {g or 'default'}: {r['decode_step_ms']} ms per token step, frac {r['decode_step_frac']} (as streamed {r['decode_step_frac_as_streamed']}), {d['value']} tokens/s whole job, chain {d['decode_chain']['launches']}/{d['decode_chain']['declined']}")
except Exception as e:
    print(f"batch {b} groups {g}: FAILED {e}")
PY
  done
done
