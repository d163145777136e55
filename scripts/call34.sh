#!/bin/bash
# runtime knobs of the launch path at B = 1 (decode step = ~300 dependent graph kernel nodes)
out=gpurun_out/r3ag; mkdir -p $out
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced --batch 1"
for cfg in "base|" "optflush0|AMD_OPT_FLUSH=0" "optflush1|AMD_OPT_FLUSH=1" "pktcap0|DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "pktcap1|DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "devkernarg0|HIP_FORCE_DEV_KERNARG=0" "devkernarg1|HIP_FORCE_DEV_KERNARG=1" "sysscope0|ROC_SYSTEM_SCOPE_SIGNAL=0" "gbatch1|DEBUG_HIP_GRAPH_BATCH_SIZE=1" "gbatch1024|DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "activewait|ROC_ACTIVE_WAIT_TIMEOUT=1000" "fgs|ROC_USE_FGS_KERNARG=0" "skipcopy|ROC_SKIP_KERNEL_ARG_COPY=1" "base2|"; do
  IFS='|' read name envs <<< "$cfg"
  env $envs timeout 300 python bench.py $common > $out/bench_$name.json 2> $out/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$name", d["value"], "tok/s; decode step", r.get("decode_step_ms"), "loop", r.get("decode_loop_ms"))
except Exception as e: print("$name failed", e)
PY
done
