#!/bin/bash
# Same-box A/B of the batch-1 / batch-2 token step: this tree against a copy of round 4's final tree under _r4/ (git archive 1031db8, built).
for round in 1 2 3; do
  for b in ${BATCHES:-1 2}; do
    for tree in _r4 .; do
      line=$(cd $tree && python bench.py --batch $b --steps 5 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced 2>/dev/null | tail -1)
      echo "round $round batch $b tree $tree: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['roofline']['decode_step_ms'], 'ms per token,', d['value'], 'tokens/s whole job, encoder', d['roofline'].get('encoder',{}).get('ms'))" "$line")"
    done
  done
done
