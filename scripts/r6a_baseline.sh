#!/bin/bash
# Round 6, first call: where the 9-32-row groups stand.  Token step (ms) per form, and the launch-by-launch timeline of ONE group of 16 / 32 rows.
#   bash scripts/r6a_baseline.sh   (writes gpurun_out/r6a_*)
export TMPDIR=/tmp
R=$PWD
run() {  # tag, env..., -- bench args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic --length-dist forced --encoder-cus 0 > gpurun_out/r6a_$tag.json 2> gpurun_out/r6a_$tag.err
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r6a_{tag}.json").read().strip().split("\n")[-1])
    r = d["roofline"]
    print(tag, r.get("decode_step_ms"), "ms per token step;", d["value"], "tokens/s whole job; kernel:", r["kernel"][:40], "launch", r.get("avg_launch_ms"))
except Exception as e:
    print(tag, "FAILED", e)
PY
  grep -i "gave up\|declin\|error" gpurun_out/r6a_$tag.err | head -3
}
for b in 8 9 12 16 24 32; do
  run b${b}_default -- --batch $b
  run b${b}_g1 -- --batch $b --groups 1
  if [ $b -gt 16 ]; then run b${b}_g1_small32 WM_LAB=1 WM_SMALL_PATH=32 -- --batch $b --groups 1; fi
done
run b64_g2_small32 WM_LAB=1 WM_SMALL_PATH=32 -- --batch 64
for cfg in "16 " "32 WM_LAB=1 WM_SMALL_PATH=32"; do
  set -- $cfg; b=$1; shift
  d=$R/gpurun_out/prof_r6a_b$b; mkdir -p $d
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --batch $b --groups 1 --steps 3 --warmup 1 --encoder-cus 0 --length-dist forced --no-roofline --no-cpu-baseline > /dev/null 2>&1
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  python scripts/step_timeline.py $t > gpurun_out/r6a_step_timeline_b${b}_g1.txt
  rm -f $t
  cat gpurun_out/r6a_step_timeline_b${b}_g1.txt
done
