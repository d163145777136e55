#!/bin/bash
# The round's final artefacts on ONE box (gpurun_out/r6z_*): the driver's command, the same under --force-dist (RCCL on one rank), the batch ladder,
# the clean rocprofv3 --kernel-trace --stats of the bench command + the per-launch distribution of the cross-attention kernel.  stderr is KEPT
# (r6z_*.err): round 5 sent it to /dev/null and a give-up of the one-launch step went unseen.  The script fails if a batch that should take the
# one-launch step shows a declined or pending give-up.
export TMPDIR=/tmp
R=$PWD
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6z_bench_line.json 2> gpurun_out/r6z_bench_line.err
python bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic > gpurun_out/r6z_bench_force_dist.json 2> gpurun_out/r6z_bench_force_dist.err
for b in 1 2 3 4 5 6 7 8 12 16 24 32 64 128 256; do
  python bench.py --batch $b --steps 5 --warmup 1 --no-cpu-baseline --no-measure-traffic > gpurun_out/r6z_bench_b$b.json 2> gpurun_out/r6z_bench_b$b.err
done
mkdir -p gpurun_out/prof_r6z
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r6z -- python3 bench.py --steps 6 --warmup 2 --encoder-cus 0 --length-dist forced --no-roofline --no-cpu-baseline > gpurun_out/r6z_bench_under_rocprof.json 2> gpurun_out/r6z_bench_under_rocprof.err
f=$(find gpurun_out/prof_r6z -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r6z_bench_kernel_stats.csv
t=$(find gpurun_out/prof_r6z -name "*kernel_trace.csv" | head -1); python scripts/trace_kernel_hist.py $t > gpurun_out/r6z_cross_attn_trace_hist.txt; rm -f $t
for b in 1 16; do
mkdir -p gpurun_out/prof_r6z_b$b
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r6z_b$b -- python3 bench.py --batch $b --steps 5 --warmup 1 --encoder-cus 0 --length-dist forced --no-roofline --no-cpu-baseline > /dev/null 2> gpurun_out/r6z_b${b}_under_rocprof.err
f=$(find gpurun_out/prof_r6z_b$b -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r6z_b${b}_kernel_stats.csv; find gpurun_out/prof_r6z_b$b -name "*kernel_trace.csv" -delete
done
python - <<'PY'
import json, sys
def line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
d = line("gpurun_out/r6z_bench_line.json")
r = d["roofline"]
print("headline", d["value"], "tokens/s", d["ms_per_step"], "ms/step")
print({k: r.get(k) for k in ("frac", "frac_best_case", "avg_launch_ms", "rocprof_avg_launch_ms", "rocprof_alone_launch_ms", "rocprof_source", "decode_step_ms", "decode_step_frac", "decode_step_frac_as_streamed", "traffic", "traffic_source")})
print("encoder", r.get("encoder"))
s = d["second_figure"]; print({k: s[k] for k in ("ms_per_batch", "ms_per_batch_pipelined", "useful_tokens_per_s", "useful_tokens_per_s_pipelined", "pipelined_encoder_released_at_layer")})
print("pipeline", {k: v for k, v in d["pipeline"].items() if k != "note"})
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "thread_sweep", "tiny_en")}, "wer", d.get("wer"))
print("| B | groups | ms per token step | decode_step_frac (as streamed) | tokens/s whole job | first token after encoder, ms | traffic / algorithmic |")
print("|---|---|---|---|---|---|---|")
bad = 0
for b in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 64, 128, 256):
    x = line(f"gpurun_out/r6z_bench_b{b}.json"); rr = x["roofline"]; ch = x["decode_chain"]
    tr = rr.get("traffic")
    ratio = f"{tr / rr['decode_step_bytes']['total']:.3f}" if (tr and b <= 8) else (f"{tr / rr['algorithmic_bytes_per_launch']:.4f} (K/V launch)" if tr and rr.get("algorithmic_bytes_per_launch") else "--")
    print(f"| {b} | {rr['decode_step_bytes']['utterance_groups']} | {rr['decode_step_ms']} | {rr['decode_step_frac']} ({rr['decode_step_frac_as_streamed']}) | {x['value']} | {x['pipeline'].get('first_token_after_encoder_ms')} | {ratio} |")
    if b <= 8 and (ch["declined"] or ch["error_pending"] or ch["launches"] == 0):
        print("   !!! batch", b, "did not run on the one-launch step:", ch); bad += 1
    if b > 8 and (ch["declined"] or ch["error_pending"]):
        print("   !!! batch", b, "chain status:", ch); bad += 1
print("| 576 | %d | %s | %s (%s) | %s | %s | -- |" % (r["decode_step_bytes"]["utterance_groups"], r["decode_step_ms"], r["decode_step_frac"], r["decode_step_frac_as_streamed"], d["value"], d["pipeline"].get("first_token_after_encoder_ms")))
f = line("gpurun_out/r6z_bench_force_dist.json"); print("force-dist", f["value"], f["n_gpus"], f["inputs"], f["gathered"])
sys.exit(1 if bad else 0)
PY
rc=$?
head -4 gpurun_out/r6z_bench_kernel_stats.csv; cat gpurun_out/r6z_cross_attn_trace_hist.txt
grep -il "gave up\|declin" gpurun_out/r6z_*.err
exit $rc
