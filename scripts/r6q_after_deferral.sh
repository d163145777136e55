#!/bin/bash
# after the deferral of small-batch prefetches: the tests that touch prefetch(), then bench.py --batch 5 eight times and the small-batch ladder
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round6.py tests/test_gpu_round2.py tests/test_gpu_round5.py -q -x 2>&1 | tail -3
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --batch 5 --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic --no-roofline > /tmp/b5.json 2> /tmp/b5.err
  python - "$i" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/b5.json").read().splitlines() if l.startswith("{")][-1])
s = d["second_figure"]; c = d["decode_chain"]
print(f"run {sys.argv[1]} batch 5: {d['ms_per_step']} ms per step ({d['value']} tokens/s); second figure {s['ms_per_batch']} / pipelined {s['ms_per_batch_pipelined']} ms; chain declined {c['declined']} launches {c['launches']}")
PY
  grep -i "gave up" /tmp/b5.err | head -1
done
for b in 1 2 4 8; do
  python bench.py --batch $b --steps 5 --warmup 1 --no-cpu-baseline --no-measure-traffic --no-roofline --length-dist forced 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pipeline']
print('batch $b', d['value'], 'tokens/s', d['ms_per_step'], 'ms per step; collect wait', p['collect_wait_ms'], 'loop', p['prefill_and_decode_loop_ms'], 'first token', p['first_token_after_encoder_ms'], 'chain', d['decode_chain']['declined'])"
done
