#!/bin/bash
# The round's final artefacts on ONE box (gpurun_out/r5z_*): the driver's command, the same under --force-dist (RCCL on one rank), the clean
# rocprofv3 --kernel-trace --stats of the bench command + the per-launch distribution of the cross-attention kernel, small batches.
export TMPDIR=/tmp
R=$PWD
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5z_bench_line.json 2> gpurun_out/r5z_bench_line.err
python bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic > gpurun_out/r5z_bench_force_dist.json 2> gpurun_out/r5z_bench_force_dist.err
for b in 1 2 3 4 5 6 7 8 16 32; do python bench.py --batch $b --steps 5 --warmup 1 --no-cpu-baseline --no-measure-traffic > gpurun_out/r5z_bench_b$b.json 2>/dev/null; done
mkdir -p gpurun_out/prof_r5z
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5z -- python3 bench.py --steps 6 --warmup 2 --encoder-cus 0 --length-dist forced --no-roofline --no-cpu-baseline > gpurun_out/r5z_bench_under_rocprof.json 2> /dev/null
f=$(find gpurun_out/prof_r5z -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r5z_bench_kernel_stats.csv
t=$(find gpurun_out/prof_r5z -name "*kernel_trace.csv" | head -1); python scripts/trace_kernel_hist.py $t > gpurun_out/r5z_cross_attn_trace_hist.txt; rm -f $t
mkdir -p gpurun_out/prof_r5z_b1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5z_b1 -- python3 bench.py --batch 1 --steps 5 --warmup 1 --encoder-cus 0 --length-dist forced --no-roofline --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/prof_r5z_b1 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r5z_b1_kernel_stats.csv; find gpurun_out/prof_r5z_b1 -name "*kernel_trace.csv" -delete
mkdir -p gpurun_out/prof_r5z_b2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5z_b2 -- python3 bench.py --batch 2 --steps 5 --warmup 1 --encoder-cus 0 --length-dist forced --no-roofline --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/prof_r5z_b2 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r5z_b2_kernel_stats.csv; find gpurun_out/prof_r5z_b2 -name "*kernel_trace.csv" -delete
for b in 4 8; do
mkdir -p gpurun_out/prof_r5z_b$b
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5z_b$b -- python3 bench.py --batch $b --steps 5 --warmup 1 --encoder-cus 0 --length-dist forced --no-roofline --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/prof_r5z_b$b -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r5z_b${b}_kernel_stats.csv; find gpurun_out/prof_r5z_b$b -name "*kernel_trace.csv" -delete
done
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5z_bench_line.json"))
r = d["roofline"]
print("headline", d["value"], "tokens/s", d["ms_per_step"], "ms/step")
print({k: r.get(k) for k in ("frac", "frac_best_case", "avg_launch_ms", "rocprof_avg_launch_ms", "rocprof_alone_launch_ms", "rocprof_source", "decode_step_ms", "decode_step_frac", "traffic")})
print("encoder", r.get("encoder"))
s = d["second_figure"]; print({k: s[k] for k in ("ms_per_batch", "ms_per_batch_pipelined", "useful_tokens_per_s", "useful_tokens_per_s_pipelined", "pipelined_encoder_released_at_layer")})
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "thread_sweep", "tiny_en")})
for b in (1, 2, 3, 4, 5, 6, 7, 8, 16, 32):
    x = json.load(open(f"gpurun_out/r5z_bench_b{b}.json")); print("batch", b, x["roofline"]["decode_step_ms"], "ms per token,", x["value"], "tokens/s")
f = json.loads(open("gpurun_out/r5z_bench_force_dist.json").readline()); print("force-dist", f["value"], f["n_gpus"])
PY
head -4 gpurun_out/r5z_bench_kernel_stats.csv; cat gpurun_out/r5z_cross_attn_trace_hist.txt
