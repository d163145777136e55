#!/bin/bash
# Which decode form for groups of 17-39 rows?  bench.py's token step (ms) at B = 2 x rows with the split-K chain (default there), the
# fused small-batch kernels stretched to 32 rows (WM_SMALL_PATH=32) and the row-split kernels pulled down (WM_ROWS_MIN=17), interleaved.
#   bash scripts/ab_decode_forms.sh > profiles/r5i_decode_forms_17_39_rows.txt
for round in 1 2 3; do
for b in 40 48 64 72; do
  for form in "default" "WM_SMALL_PATH=32" "WM_ROWS_MIN=17"; do
    if [ "$form" = "default" ]; then envs=""; else envs="WM_LAB=1 $form"; fi
    line=$(env $envs python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --encoder-cus 0 2>/dev/null | tail -1)
    echo "round $round batch $b ($((b/2)) rows per group) $form: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['roofline']['decode_step_ms'], 'ms per token step,', d['value'], 'tokens/s whole job')" "$line")"
  done
done
done
