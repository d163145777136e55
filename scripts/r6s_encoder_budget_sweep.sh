#!/bin/bash
# round 6: the CU budget of the prefetched encoder pass by batch size (bench.py --encoder-cus; 0 = one stage after the other)
export TMPDIR=/tmp
for b in ${BATCHES:-12 16 32 64 128 256}; do
  for cus in 0 48 96 144 192; do
    line=$(python bench.py --batch $b --encoder-cus $cus --steps 4 --warmup 2 --no-cpu-baseline --no-measure-traffic --no-roofline --length-dist forced 2>/dev/null | grep '^{' | tail -1)
    python - "$b" "$cus" "$line" <<'PY'
import json, sys
b, cus, line = sys.argv[1:4]
d = json.loads(line); p = d["pipeline"]
print(f"batch {b} encoder CUs {cus}: {d['value']} tokens/s, {d['ms_per_step']} ms per step; loop {p['prefill_and_decode_loop_ms']} ms, collect wait {p['collect_wait_ms']}, encoder in the open {p['encoder_in_the_open_ms']}, prefetched pass {p['encoder_prefetch_ms']}")
PY
  done
done
