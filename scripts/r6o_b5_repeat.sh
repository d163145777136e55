#!/bin/bash
# how often does the one-launch step give up in bench.py --batch 5 (profiles/r6z_bench_b5.err had one)?  alternating graph_prefill on / off
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  for gp in 1 0; do
    WM_GRAPH_PREFILL=$gp python bench.py --batch 5 --steps 5 --warmup 1 --no-cpu-baseline --no-measure-traffic > /tmp/b5.json 2> /tmp/b5.err
    python - "$i" "$gp" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/b5.json").read().splitlines() if l.startswith("{")][-1])
s = d["second_figure"]; c = d["decode_chain"]
print(f"run {sys.argv[1]} graph_prefill {sys.argv[2]}: {d['ms_per_step']} ms per step; second figure {s['ms_per_batch']} / pipelined {s['ms_per_batch_pipelined']} ms; chain declined {c['declined']} launches {c['launches']}")
PY
    grep -i "gave up" /tmp/b5.err | head -1
  done
done
