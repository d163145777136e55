#!/bin/bash
# per-launch durations and gaps of the decode step at small batches (launch-per-kernel path): BATCHES="4 8" bash scripts/prof_small_batch.sh <tag>
export TMPDIR=/tmp
R=$PWD; tag=${1:-r5ah}
for b in ${BATCHES:-4 8}; do
  d=$R/gpurun_out/prof_${tag}_b$b; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --batch $b --steps 3 --warmup 1 --encoder-cus 0 --length-dist forced --no-roofline --no-cpu-baseline > /dev/null 2>&1
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  python scripts/step_timeline.py $t > gpurun_out/${tag}_step_timeline_b$b.txt
  rm -f $t
  python bench.py --batch $b --steps 5 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('unprofiled: batch', $b, d['roofline']['decode_step_ms'], 'ms per token')" >> gpurun_out/${tag}_step_timeline_b$b.txt
  cat gpurun_out/${tag}_step_timeline_b$b.txt
done
