#!/bin/bash
# packet-capture default set by the package at import: does it take effect when torch was imported first?  + the whole GPU suite
out=gpurun_out/r3ai; mkdir -p $out
common="--steps 2 --warmup 1 --no-cpu-baseline --no-measure-traffic --encoder-cus 0 --length-dist forced --batch 1"
timeout 300 python bench.py $common > $out/bench_b1_default.json 2> $out/bench_b1_default.err
timeout 300 python -c "import torch, runpy, sys; sys.argv=['bench.py']+'$common'.split(); runpy.run_path('bench.py', run_name='__main__')" > $out/bench_b1_torch_first.json 2> $out/bench_b1_torch_first.err
timeout 300 python -c "import torch, runpy, sys; torch.cuda.is_available(); sys.argv=['bench.py']+'$common'.split(); runpy.run_path('bench.py', run_name='__main__')" > $out/bench_b1_cuda_first.json 2> $out/bench_b1_cuda_first.err
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout 300 python bench.py $common > $out/bench_b1_cap1.json 2> $out/bench_b1_cap1.err
for n in b1_default b1_torch_first b1_cuda_first b1_cap1; do python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$n.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print("$n", d["value"], "tok/s; decode step", r.get("decode_step_ms"), d.get("hip_runtime_knobs"))
except Exception as e: print("$n failed", e)
PY
done
timeout 2400 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
