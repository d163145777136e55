#!/bin/bash
# round 6: the refactored bench.py on the box -- a small batch on the one-launch step, 16 utterances as two stream-parallel groups (no chain launch, nothing declined), RCCL on one rank
export TMPDIR=/tmp
for cfg in "b4 --batch 4" "b16 --batch 16" "b8_force_dist --batch 8 --force-dist" "b8_scatter --batch 8 --force-dist --scatter-inputs"; do
  set -- $cfg; tag=$1; shift
  python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic > gpurun_out/r6c_$tag.json 2> gpurun_out/r6c_$tag.err
  python - $tag <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r6c_{sys.argv[1]}.json").read().strip().split("\n")[-1])
r = d["roofline"]
print(sys.argv[1], d["value"], "tokens/s;", r["decode_step_ms"], "ms per token; frac", r["decode_step_frac"], "as streamed", r["decode_step_frac_as_streamed"], "bytes", r["decode_step_bytes"]["total"], r["decode_step_bytes"]["weight_reread_factor"])
print("   chain", d["decode_chain"], "inputs", d["inputs"], "affinity", d["affinity"], "gathered", d["gathered"])
PY
  tail -2 gpurun_out/r6c_$tag.err
done
