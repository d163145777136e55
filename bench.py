"""bench.py -- throughput of the Whisper hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B --decode-steps T --config int8 ...]
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One STEP = one pass of the hot path over one batch of B synthetic 30 s log-mel spectrograms per
GPU, already resident in HBM: encoder -> cross K/V projection -> language-ID pass (1 token) ->
prefill (3 tokens) -> T forced greedy decode steps with Whisper's logit rules (EOT is ignored so
that random weights decode exactly T tokens), then the gather of the token ids.  Everything the
reference's run.py times per utterance (W/run.py:56-61) is inside the timed region.

Consecutive steps are software-pipelined the way a transcription job over many batches is (summarize.py
--overlap_encoder): while the decode loop of step n runs, the encoder of step n + 1 runs beside it on --encoder-cus CUs
(default 96; 0 = one stage after the other).  All K encoders, projections, language passes, prefills and decode loops of
the K timed steps run inside the timed region: the first encoder in the open, the last decode loop with nothing beside it.

Workload = BASELINE.json configs[3]: Whisper large-v2, weight-only int8 + int8 KV cache + fp16
cross K/V ("the configuration the metric is quoted on"); random-init weights (no checkpoint exists
on any box), KV scales calibrated with the reference's rule (torch_whisper_convert.py -kv).

Prints ONE JSON line (rank 0): value = decoded tokens per second over the whole job
(n_gpus * B * T tokens per step / step time), plus `rtf`, `roofline` for the dominant kernel (decode
cross-attention, HBM-bound: measured in situ with HIP events on its launch stream) and
`cpu_baseline` (the oracle, a port of the reference's PyTorch path, timed on this box's host cores: one full-depth clip).
`python bench.py --gpus N` without a launcher starts the N ranks itself (fresh child processes).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import shutil
import sys
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "eddie-wang-hackathon2023_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import native  # noqa: E402,F401  (first: sets the HIP runtime's graph-replay default before anything initialises the runtime)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {"fp16": (False, False), "int8wo": (True, False), "int8kv": (False, True), "int8": (True, True),
           "int4": ("int4", True),       # weight-only precision (False / True = int8 / "int4"), int8 KV cache
           "int8x": (True, True)}        # "int8" + int8 cross-attention K/V: an opt-in mode BEYOND the reference
WORKLOADS = {"fp16": "fp16 GEMMs + fp16 self-KV + fp16 cross-KV", "int8wo": "weight-only int8 GEMMs + fp16 self-KV + fp16 cross-KV",
             "int8kv": "fp16 GEMMs + int8 self-KV + fp16 cross-KV", "int8": "weight-only int8 GEMMs + int8 self-KV + fp16 cross-KV",
             "int4": "weight-only int4 GEMMs + int8 self-KV + fp16 cross-KV",
             "int8x": "weight-only int8 GEMMs + int8 self-KV + INT8 cross-KV (opt-in, beyond the reference's numerics)"}
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
MFMA_PEAK_TFLOPS = 2500.0    # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline figure includes 2:1 sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=576, help="utterances per GPU per step")
    ap.add_argument("--decode-steps", type=int, default=128, help="forced greedy tokens per utterance")
    ap.add_argument("--model", type=str, default="large-v2")
    ap.add_argument("--config", type=str, default="int8", choices=list(CONFIGS))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--engine-cache", type=str, default="/tmp/wm_bench_engines")
    ap.add_argument("--encoder-cus", type=int, default=96,
                    help="CUs the next step's encoder runs on beside the current step's decode loop (0: no pipelining, one stage after the other)")
    ap.add_argument("--groups", type=int, default=0, help="utterance groups of the decode loop (0: the library's choice)")
    ap.add_argument("--length-dist", type=str, default="librispeech-like", choices=["forced", "librispeech-like"],
                    help="SECOND figure beside the headline (which always decodes --decode-steps forced tokens per utterance): "
                         "extra untimed-for-the-headline steps in which utterances end after LibriSpeech-like token counts "
                         "(per-row completion: finished rows drop out of the attention kernels); forced = skip it")
    ap.add_argument("--no-measure-traffic", action="store_true",
                    help="do not run the rocprofv3 --pmc FETCH_SIZE pass of the dominant kernel (roofline.traffic then comes from the committed pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager-loop", action="store_true", help="lab: the decode loop with eager launches instead of replayed graphs (A/B runs of launch-level experiments)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank (self-test)")
    ap.add_argument("--scatter-inputs", action="store_true",
                    help="rank 0 draws the GLOBAL batch and scatters it over RCCL (the round-1..5 input path: 2.2 GB at 8 x 576 clips); default: every "
                         "rank draws its own shard from (seed, rank).  The time either takes is reported (inputs.ms), outside the timed region")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the ranks of a multi-rank job to their GPU's NUMA cores (dp.pin_rank_to_gpu_numa)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="tests/test_dp_gloo.py: the whole control flow of main() -- barriers, input shards, timed steps, gather, per-rank times, the one "
                         "JSON line -- on CPU over gloo with stand-in engines (no GPU, no HIP library); never a measurement")
    return ap.parse_args()


def build_engines(args, out_dir: Path):
    """Random-init checkpoint -> (calibration) -> engine directory, all on the GPU of this rank."""
    import build as B
    import synthetic
    import torch_whisper_convert as TWC
    wo, i8kv = CONFIGS[args.config]
    t0 = time.time()
    ck = synthetic.synthetic_checkpoint(args.model, args.seed, device="cuda")
    argv = ["--output_dir", str(out_dir), "--log_level", "error", "--use_gpt_attention_plugin", "--use_gemm_plugin",
            "--use_layernorm_plugin"] + (["--use_weight_only"] if wo else []) + (["--weight_only_precision", "int4"] if wo == "int4" else [])
    if i8kv:
        # int8-KV calibration.  The reference calibrates the un-quantised fp16 model; to keep
        # the bench start-up short this uses the engines of the same weight precision with an fp16 cache.
        calib = Path(str(out_dir) + "_calib")
        B.build_from_checkpoint(ck, B.parse_arguments(["--output_dir", str(calib)] + argv[2:]))
        mels = synthetic.synthetic_mel(4, 2 * ck["dims"]["n_audio_ctx"], ck["dims"]["n_mels"], 4321)
        amax = TWC.capture_kv_activation_range(calib, mels, batch=4, sample_len=16, ignore_eot=True)
        qdir = TWC.write_kv_scales(str(out_dir) + "_quantize", amax, {"source": "bench.py synthetic calibration"})
        if args.config == "int8x":
            TWC.write_cross_kv_scales(str(out_dir) + "_quantize", TWC.capture_cross_kv_range(calib, mels, batch=4))
            argv += ["--int8_cross_kv"]
        shutil.rmtree(calib, ignore_errors=True)
        argv += ["--int8_kv_cache", "--quantize_dir", str(qdir)]
    B.build_from_checkpoint(ck, B.parse_arguments(argv))
    del ck
    torch.cuda.empty_cache()
    return time.time() - t0


def cpu_baseline(args, decode_steps: int, time_cap_s: float = 150.0):
    """The oracle (kind "port": our CPU restatement of the reference's PyTorch path, pinned to the reference by
    tests/golden) timed on the host cores: ONE clip through the whole path at FULL depth in the reference's fp16-input
    mode (fp32-stored parameters, fp16 activations, W/torch_model.py:25-45 -- how W/summarize.py:81-84,121 runs it) --
    encoder, cross K/V, language-ID pass, 3-token prefill and the greedy steps.  The decode loop is timed for as many
    of the `decode_steps` tokens as fit in the time cap (every step costs the same: the per-token time of the measured
    steps is applied to the rest, and `sample` says how many were measured).
    Threads: a short sweep first (encoder once + 3 greedy steps at 8 / 16 / 32 / 64 / 128 threads, as far as the box has cores) picks
    the count with the fastest whole clip (127 steps + encoder); the sweep is reported (`thread_sweep`), `cores` is what the timed
    clip used.  `tiny_en`: the same path for tiny.en's shape (BASELINE.json configs[0]), every step measured."""
    from oracle.whisper_oracle import Dims, OracleConfig, OracleModel
    import synthetic
    n_cpu = os.cpu_count() or 1

    def build(model_name):
        d = dict(synthetic.DIMS[model_name])
        # same distributions as the engines' weights, drawn on the GPU (seconds instead of minutes for 1.5e9 values), moved to the host
        sd = synthetic.synthetic_state_dict(d, args.seed, device="cuda" if torch.cuda.is_available() else None)
        dims = Dims(**d)
        model = OracleModel(dims, {k: v.cpu() for k, v in sd.items()}, OracleConfig(act="float16"))
        return dims, model, synthetic.synthetic_mel(1, 2 * dims.n_audio_ctx, dims.n_mels, 1234)

    def clip(dims, model, mel, steps, cap_s):
        t_start = time.perf_counter()
        with torch.no_grad():
            t = time.perf_counter(); xa = model.encoder(mel); t_enc = time.perf_counter() - t
            t = time.perf_counter(); ckv = model.cross_kv(xa); t_ckv = time.perf_counter() - t
            sot = dims.n_vocab - 1607                          # <|startoftranscript|> of either vocabulary
            t = time.perf_counter(); model.decoder(torch.tensor([[sot]]), ckv, None); t_lang = time.perf_counter() - t
            t = time.perf_counter(); logits, kv = model.decoder(torch.tensor([[sot, sot + 1, sot + 101]]), ckv, None); t_pre = time.perf_counter() - t
            n_meas, t_loop = 0, 0.0
            while n_meas < steps - 1 and (n_meas < 3 or time.perf_counter() - t_start < cap_s):
                t = time.perf_counter()
                logits, kv = model.decoder(logits[:, -1:].argmax(-1), ckv, kv)
                t_loop += time.perf_counter() - t
                n_meas += 1
        return dict(enc=t_enc, ckv=t_ckv, lang=t_lang, pre=t_pre, step=t_loop / max(n_meas, 1), n=n_meas)

    dims, model, mel = build(args.model)
    sweep = {}
    for n in (8, 16, 32, 64, 128):
        if n <= n_cpu:
            torch.set_num_threads(n)
            r = clip(dims, model, mel, 4, 0.0)
            sweep[str(n)] = {"encoder_s": round(r["enc"], 3), "step_s": round(r["step"], 4),
                             "clip_estimate_s": round(r["enc"] + r["ckv"] + r["lang"] + r["pre"] + (decode_steps - 1) * r["step"], 2)}
    cores = int(min(sweep, key=lambda k: sweep[k]["clip_estimate_s"])) if sweep else min(n_cpu, 32)
    torch.set_num_threads(cores)
    r = clip(dims, model, mel, decode_steps, time_cap_s)
    total = r["enc"] + r["ckv"] + r["lang"] + r["pre"] + (decode_steps - 1) * r["step"]
    out = {
        "value": round(decode_steps / total, 3), "unit": "tokens/s", "cores": cores, "host_cpu_count": n_cpu, "kind": "port",
        "rtf": round(total / 30.0, 3),
        "sample": (f"oracle, fp16-input mode, batch 1, {args.model} at full depth ({dims.n_audio_layer}+{dims.n_text_layer} layers): "
                   f"encoder {r['enc']:.2f}s, cross-K/V {r['ckv']:.2f}s, language-ID pass {r['lang']:.2f}s, prefill {r['pre']:.2f}s, "
                   f"{r['n']} of {decode_steps - 1} greedy steps measured at {r['step']:.3f}s each"
                   + ("" if r["n"] == decode_steps - 1 else f" (time cap {time_cap_s:.0f}s; the rest priced at that rate)")),
        "thread_sweep": sweep or None,
        "thread_sweep_note": "encoder once + 3 greedy steps per thread count (torch.set_num_threads), whole clip priced from them; `cores` = the fastest",
    }
    del model
    if "tiny.en" in synthetic.DIMS and args.model != "tiny.en":
        d2, m2, mel2 = build("tiny.en")
        r2 = clip(d2, m2, mel2, decode_steps, 60.0)
        t2 = r2["enc"] + r2["ckv"] + r2["lang"] + r2["pre"] + (decode_steps - 1) * r2["step"]
        out["tiny_en"] = {"value": round(decode_steps / t2, 2), "unit": "tokens/s", "rtf": round(t2 / 30.0, 4), "cores": cores,
                          "sample": (f"oracle, fp16-input mode, batch 1, tiny.en's shape ({d2.n_audio_layer}+{d2.n_text_layer} layers, {d2.n_vocab} tokens; "
                                     f"BASELINE.json configs[0]): encoder {r2['enc']:.3f}s, cross-K/V {r2['ckv']:.3f}s, first pass {r2['lang']:.3f}s, "
                                     f"prefill {r2['pre']:.3f}s, {r2['n']} greedy steps at {r2['step'] * 1e3:.1f} ms each")}
    return out


def wer_block(args) -> dict:
    """BASELINE.json's third metric: WER of the benchmarked configuration against the fp16 engines on LibriSpeech test-clean, through
    summarize.py (the reference's own evaluation, W/summarize.py:72-181: clips over 30 s skipped, English normaliser, corpus WER).
    Needs real weights and real audio: WM_CHECKPOINT = an OpenAI `.pt` ({'dims', 'model_state_dict'}: W/build.py:146-147) and
    WM_LIBRISPEECH = a test-clean directory (FLAC + *.trans.txt).  No box of this project has had either, so the block says
    "not measured" and why -- it never invents a number.  With both: fp16 engines and the benchmarked config are built from the
    checkpoint (int8 KV scales calibrated on the first utterance of each chapter, the reference's calibration set, W/trans_data.py:20-38),
    both transcribe the set, the block carries both WERs and their difference (north_star: within 0.1)."""
    ck, ds = os.environ.get("WM_CHECKPOINT"), os.environ.get("WM_LIBRISPEECH")
    if not ck or not ds or not os.path.exists(ck) or not os.path.isdir(ds):
        return {"wer": "not measured",
                "wer_note": "needs real weights and audio: set WM_CHECKPOINT (large-v2.pt) and WM_LIBRISPEECH (test-clean directory); "
                       f"WM_CHECKPOINT {'missing' if not ck or not os.path.exists(ck) else 'present'}, "
                       f"WM_LIBRISPEECH {'missing' if not ds or not os.path.isdir(ds) else 'present'} (SURVEY 8d: otherwise report not measured)"}
    try:
        import tempfile
        import build as B
        import summarize as S
        import torch_whisper_convert as TWC
        wo, i8kv = CONFIGS[args.config]
        model = torch.load(ck, map_location="cpu")
        out = {}
        with tempfile.TemporaryDirectory(prefix="wm_wer_") as tmp:
            base = ["--log_level", "error", "--use_gpt_attention_plugin", "--use_gemm_plugin", "--use_layernorm_plugin"]
            fp16_dir = os.path.join(tmp, "fp16")
            B.build_from_checkpoint(model, B.parse_arguments(["--output_dir", fp16_dir] + base))
            cfg_dir = os.path.join(tmp, args.config)
            argv = ["--output_dir", cfg_dir] + base + (["--use_weight_only"] if wo else []) + (["--weight_only_precision", "int4"] if wo == "int4" else [])
            if i8kv:
                pairs = S.discover(ds)
                first_of_chapter = {}
                for f, _ in pairs:
                    first_of_chapter.setdefault(f.parent, f)
                audio = [S.load_audio(str(f)) for f in sorted(first_of_chapter.values())[:87]]
                mels = S.mel_batch(audio, torch.device("cuda"))
                amax = TWC.capture_kv_activation_range(fp16_dir, mels, batch=8)
                argv += ["--int8_kv_cache", "--quantize_dir", str(TWC.write_kv_scales(os.path.join(tmp, "quantize"), amax, {"source": "bench.py wer_block"}))]
            B.build_from_checkpoint(model, B.parse_arguments(argv))
            del model
            for name, d in (("fp16", fp16_dir), (args.config, cfg_dir)):
                rep = S.main(S.parse_arguments(["--test_trt_llm", "--engine_dir", d, "--dataset_dir", ds, "--batch_size", "64", "--log_level", "error"]))
                r = rep["whisper-mi355"]
                out[name] = {"wer_percent": round(100 * r["wer"], 3), "utterances": r["utterances"], "seconds": round(r["seconds"], 1)}
        out["delta_percent_points"] = round(out[args.config]["wer_percent"] - out["fp16"]["wer_percent"], 3)
        out["reference_published"] = "README.md:166-173: PyTorch fp16 4.19, TRT-LLM 3.91, int8 weight-only 2.76, int8 KV 4.32 (A10, other hardware)"
        return {"wer": out}
    except Exception as e:       # noqa: BLE001 -- an evaluation problem must not cost the bench line
        return {"wer": "not measured", "wer_note": f"WM_CHECKPOINT / WM_LIBRISPEECH are set but the evaluation failed: {type(e).__name__}: {e}"}


def pmc_traffic(group: int, kv_bytes: int) -> dict:
    """`roofline.traffic`: HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE pass whose summary is committed under
    profiles/ (newest profiles/*_pmc_cross_attn.json: {"fetch_bytes_per_utterance_layer": ..., "source": ...})."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_cross_attn.json")))
    if not files or kv_bytes != 2:
        return {"traffic": None, "traffic_source": None}
    with open(files[-1]) as f:
        rec = json.load(f)
    return {"traffic": int(round(group * rec["fetch_bytes_per_utterance_layer"])), "traffic_source": os.path.relpath(files[-1], ROOT)}


def measure_traffic(group: int, kv_bytes: int, timeout_s: float = 240.0) -> dict:
    """`roofline.traffic` measured in this run: a rocprofv3 --pmc FETCH_SIZE pass (its own process, counters only, as
    MI355X_MICROARCH.md prescribes) over scripts/cross_attn_probe.py at this run's launch shape -- the same kernel, grid and
    K/V layout the decode loop launches, four K/V buffers rotated so that the Infinity Cache cannot serve re-reads.
    HBM bytes per launch = FETCH_SIZE (KB) x 1024 x 2 (gfx950: the counter reports half the bytes of 16 B/lane streaming
    reads).  Returns {} when rocprofv3 is missing or the pass fails (the caller then falls back to the committed pass)."""
    import csv
    import glob
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or kv_bytes != 2:
        return {}
    out = tempfile.mkdtemp(prefix="wm_pmc_")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = [exe, "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.join(ROOT, "scripts", "cross_attn_probe.py"), str(group), "8"]
        proc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
        vals = []
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if "attn_cross_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == "FETCH_SIZE":
                        vals.append(float(row["Counter_Value"]))
        if proc.returncode != 0 or not vals:
            return {}
        fetch = float(np.mean(vals)) * 1024.0 * 2.0
        return {"traffic": int(round(fetch)),
                "traffic_source": f"measured in this run: rocprofv3 --pmc FETCH_SIZE over scripts/cross_attn_probe.py {group} 8 "
                                  f"({len(vals)} launches; FETCH_SIZE KB x 1024 x 2, the gfx950 correction for 16 B/lane streaming reads)"}
    except Exception:       # noqa: BLE001 -- a profiler problem must not cost the bench line
        return {}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def profile_round_key(path: str):
    """Sort key of a committed profile by its ROUND NAME (r4z < r5a < r5b < r5aa): file times mean nothing after a checkout."""
    import re
    m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
    return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (-1, 0, "")


def rocprof_reference(kernel_prefix: str = "attn_cross_kernel<1, false, 0") -> dict:
    """Average duration of the dominant kernel over every launch in the committed rocprofv3 --kernel-trace --stats run of the
    CURRENT round (profiles/r<N>*_bench_kernel_stats.csv with the highest round name; N = the highest round any file under
    profiles/ carries), for the reader to set beside the live HIP-event figure.  The kernel is matched by PREFIX (template
    arguments added later do not break the match).  No file of the current round -> rocprof_source: null and a warning on
    stderr -- never a silent fall-back to an older round's kernel."""
    import csv
    import glob
    every = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*")) if profile_round_key(f)[0] >= 0]
    rnd = max((profile_round_key(f)[0] for f in every), default=-1)
    files = sorted((f for f in every if f.endswith("_bench_kernel_stats.csv") and profile_round_key(f)[0] == rnd), key=profile_round_key)
    none = {"rocprof_avg_launch_ms": None, "rocprof_calls": None, "rocprof_source": None}
    if not files:
        print(f"bench.py: no profiles/r{rnd}*_bench_kernel_stats.csv of the current round: roofline.rocprof_source is null", file=sys.stderr)
        return none
    f = files[-1]
    # the per-launch distribution behind that mean, from the --kernel-trace CSV of the same command (scripts/trace_kernel_hist.py):
    # rocprofv3 does not serialise the groups' queues completely -- a few percent of the launches overlap another group's launch
    # and take twice as long, which is most of the gap between the mean over all launches and the launch alone on the chip
    alone = {}
    hists = sorted((h for h in every if h.endswith("_cross_attn_trace_hist.txt") and profile_round_key(h)[0] == rnd), key=profile_round_key)
    if hists and "attn_cross_kernel" in kernel_prefix:
        try:
            import re
            txt = open(hists[-1]).read()
            m1 = re.search(r"overlap another launch of the kernel in time: ([0-9.]+) %", txt)
            m2 = re.search(r"launches alone on the chip: mean ([0-9.]+) us, median ([0-9.]+) us", txt)
            if m1 and m2:
                alone = {"rocprof_alone_launch_ms": round(float(m2.group(1)) * 1e-3, 5), "rocprof_alone_median_ms": round(float(m2.group(2)) * 1e-3, 5),
                         "rocprof_overlapped_share": round(float(m1.group(1)) / 100.0, 4), "rocprof_trace_source": os.path.relpath(hists[-1], ROOT)}
        except Exception:       # noqa: BLE001
            alone = {}
    try:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Name") or row.get("KernelName") or ""
                if kernel_prefix in name:
                    avg_ns = float(row.get("AverageNs") or row.get("Average") or 0)
                    if avg_ns > 0:
                        return {"rocprof_avg_launch_ms": round(avg_ns * 1e-6, 5), "rocprof_calls": int(float(row.get("Calls", 0))),
                                "rocprof_source": os.path.relpath(f, ROOT), **alone}
    except Exception as e:       # noqa: BLE001
        print(f"bench.py: {os.path.relpath(f, ROOT)} could not be read ({e}): roofline.rocprof_source is null", file=sys.stderr)
        return none
    print(f"bench.py: {os.path.relpath(f, ROOT)} has no row for {kernel_prefix}...: roofline.rocprof_source is null", file=sys.stderr)
    return none


def librispeech_like_lengths(n: int, t_max: int, seed: int = 2620) -> np.ndarray:
    """Sampled-token counts per utterance for the second figure: LibriSpeech test-clean has 2 620 utterances of 1.3-35 s
    (mean 7.4 s, long right tail); a Whisper transcript of it carries about 3.6 tokens per second of speech plus the two
    timestamps.  Durations ~ log-normal(median 6.3 s, sigma 0.62) clipped to [1.3, 30] s (clips over 30 s are skipped by the
    reference, W/summarize.py:118-120), tokens = round(3.6 x duration) + 2, capped at the decode budget.  Sorted ascending:
    a transcription job batches clips by duration (summarize.py), so the utterance groups of a batch -- contiguous
    slices -- hold clips of similar length and the group of the shortest clips finishes as a whole."""
    rng = np.random.Generator(np.random.PCG64(seed))
    dur = np.clip(np.exp(rng.normal(np.log(6.3), 0.62, size=n)), 1.3, 30.0)
    tok = np.minimum(np.rint(3.6 * dur).astype(np.int64) + 2, t_max)
    return np.sort(np.maximum(tok, 1))


def encoder_flops_per_clip(d: dict) -> float:
    """Algorithmic FLOPs of one encoder pass (2.272 TFLOP per clip at large-v2): the four Linears and the attention
    of every block, the two convolutions."""
    T, C, L, M = d["n_audio_ctx"], d["n_audio_state"], d["n_audio_layer"], d["n_mels"]
    per_layer = 2.0 * T * (3 * C * C + C * C + 2 * C * 4 * C) + 4.0 * T * T * C
    conv = 2.0 * (2 * T) * M * 3 * C + 2.0 * T * C * 3 * C
    return L * per_layer + conv


def in_situ_probe(dec, lib, xa, B: int, n_micro: int, algo_bytes: int, steps: int = 20, beside=None) -> dict:
    """Durations of the cross-attention launches as the decode loop runs them: replayed from the captured graphs, the
    utterance groups' launches and short-kernel chains sharing the chip.  A 1-thread stamp kernel before and after every
    launch writes the device wall clock (wm_debug_timeline); the graphs are re-captured with the stamps for this probe
    and dropped afterwards.  `beside` = (encoder, mel, cus): the same probe with the next step's encoder running beside the
    loop on that many CUs, as in the pipelined steps."""
    import native
    st = dec._state[B]
    st["graphs"].clear()
    cap = n_micro * 2 * dec.decoder_config["num_layers"] * (steps + 4)
    buf = torch.zeros(1 + 3 * cap, dtype=torch.int64, device=xa.device)
    keep = dec.sample_len
    native.check(lib.wm_debug_timeline(buf.data_ptr(), cap))
    try:
        dec.sample_len = steps
        dec.main_loop(xa, ignore_eot=True)
        torch.cuda.synchronize()
        if beside is not None:           # graph capture synchronises the device: capture first (above), then measure beside the encoder
            buf.zero_()
            beside[0].prefetch(beside[1], beside[2])
            time.sleep(0.05)             # let the encoder's first launches reach the GPU
            dec.main_loop(xa, ignore_eot=True)
            beside[0].collect()
            torch.cuda.synchronize()
    finally:
        native.check(lib.wm_debug_timeline(None, 0))
        dec.sample_len = keep
        st["graphs"].clear()
    n = min(int(buf[0].item()) & 0xffffffff, cap)
    ev = buf[1:1 + 3 * n].view(-1, 3).cpu().numpy()
    durs, gaps, spans = [], [], []
    skip = 2 * dec.decoder_config["num_layers"]            # the eager prefill and the captured step
    for tag in sorted(set(ev[:, 0].tolist())):
        e = ev[ev[:, 0] == tag]
        e = e[np.argsort(e[:, 2], kind="stable")]
        starts, ends = e[e[:, 1] % 2 == 0][:, 2], e[e[:, 1] % 2 == 1][:, 2]
        m = min(len(starts), len(ends))
        starts, ends = starts[skip:m], ends[skip:m]
        if len(starts) < 2:
            continue
        durs.append((ends - starts) / 100.0)               # wall_clock64 ticks of 10 ns -> us
        gaps.append((starts[1:] - ends[:-1]) / 100.0)
        spans += [(int(t), 1) for t in starts] + [(int(t), -1) for t in ends]
    if not durs:
        return {}
    spans.sort()
    level, prev, acc = 0, None, {}
    for t, d in spans:
        if prev is not None:
            acc[level] = acc.get(level, 0) + (t - prev)
        level, prev = level + d, t
    tot = float(sum(acc.values())) or 1.0
    dur_us = float(np.mean(np.concatenate(durs)))
    return {"in_situ_launch_ms": round(dur_us * 1e-3, 5),
            "in_situ_frac": round(algo_bytes / (dur_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "in_situ_chain_between_launches_ms": round(float(np.mean(np.concatenate(gaps))) * 1e-3, 5),
            "in_situ_launches_in_flight": {str(k): round(v / tot, 3) for k, v in sorted(acc.items())},
            "in_situ_note": f"graph-replayed launches, {n_micro} utterance groups sharing the chip; device-clock stamps around each launch "
                            f"({int(sum(len(d) for d in durs))} launches over {steps} extra untimed decode steps)"}


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as FRESH child processes (one per GPU) through
    torch.distributed.run and relay rank 0's JSON line.  Nothing in this process has touched the GPU yet
    (torch.cuda.device_count() does not initialise it on this image), and the children are started with subprocess,
    never exec'ed from a GPU-initialised process."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node shows {n_dev} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in proc.stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if proc.returncode == 0 and lines:
        print(lines[-1], flush=True)
        return 0
    print(f"bench.py: the {args.gpus}-rank job failed (rc={proc.returncode})", file=sys.stderr)
    return proc.returncode or 1


class _HostEvent:
    """What main() needs of an event when there is no GPU (--stub-engine): a host time stamp."""
    def record(self, *_):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _GpuRuntime:
    """The few device services main()'s control flow uses, so that the same flow runs over gloo on CPU with stand-ins (_StubRuntime)."""
    backend = "nccl"

    def device(self, local_rank):
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
        torch.cuda.set_device(local_rank)
        return torch.device("cuda", local_rank)

    def sync(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def generator(self, dev, seed):
        return torch.Generator(device=dev).manual_seed(seed)

    def memory(self, dev):
        free_b, total_b = torch.cuda.mem_get_info(dev)
        return int(torch.cuda.memory_allocated(dev)), int(total_b - free_b)


class _StubRuntime(_GpuRuntime):
    backend = "gloo"

    def device(self, local_rank):
        return torch.device("cpu")

    def sync(self):
        pass

    def event(self):
        return _HostEvent()

    def memory(self, dev):
        return 0, 0


class _StubEngines:
    """Stand-ins for WhisperEncoding / WhisperDecoding with the surface main() drives (--stub-engine).  "Audio features" are the per-clip
    mean of the mel, a token row is a function of its clip alone -- so the gathered result says whether every rank's rows arrived in order."""
    class _Sess:
        class engine:
            weight_bytes = 0
    session = decoder_session = cross_attn_session = _Sess
    time_prefetch, prefetch_events, last_release_layer = False, [], None
    use_graphs, micro_batches, sample_len, initial_token_length, sample_begin = True, None, 8, 3, 3
    lang_id_sequential = groups_sequential = False

    class tokenizer:
        eot = 50257

    def __init__(self):
        self._pending = None

    # encoder side
    def get_audio_features_async(self, mel):
        return mel.float().mean(dim=(1, 2))

    def prefetch(self, mel, cus):
        self._pending = self.get_audio_features_async(mel)

    def collect(self):
        xa, self._pending = self._pending, None
        return xa

    def loop_ended(self):
        pass

    # decoder side
    def detect_language(self, xa):
        return ["en"] * xa.shape[0], None

    def _groups(self, n):
        return 1, [(0, n)]

    def balanced_order(self, n):
        return list(range(n))

    @staticmethod
    def token_rows(xa, n_tokens):
        base = (xa * 1e6).round().to(torch.int64).abs() % 40000
        return base[:, None] + torch.arange(n_tokens, dtype=torch.int64)[None, :]

    def main_loop(self, xa, ignore_eot=False, row_limit=None):
        n = xa.shape[0]
        t = self.token_rows(xa, self.initial_token_length + self.sample_len)
        if row_limit is not None:
            cut = self.initial_token_length + row_limit.to(torch.int64)[:, None]
            t = torch.where(torch.arange(t.shape[1])[None, :] >= cut, torch.full_like(t, self.tokenizer.eot), t)
        return t, -xa.float(), [0.0] * n


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a different GPU count", file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; before anything initialises the GPU
    # stdout carries ONE line, the JSON: RCCL prints a version banner on the C-level stdout when its communicator is torn down (five
    # lines BEHIND the JSON line in profiles/r5z_bench_force_dist.json -- a driver that reads the last line of stdout would have read
    # "Librccl path : ..."), and any other library may do the like.  File descriptor 1 is pointed at stderr for the life of the process;
    # the line goes to the saved descriptor.  (Not under --stub-engine: the test captures sys.stdout.)
    json_fd = None
    if not args.stub_engine:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    import dp
    rt = _StubRuntime() if args.stub_engine else _GpuRuntime()
    if args.stub_engine:            # the control-flow test: nothing below may touch a GPU, the probes are off
        args.no_roofline = args.no_cpu_baseline = args.no_measure_traffic = True
    # host placement BEFORE the first GPU call (the runtime's and torch's threads inherit the mask): every rank of a multi-rank job on the
    # CPUs next to its GPU, ranks that share a NUMA node on disjoint slices of it.  One rank alone is left where the OS put it (nothing to
    # keep apart -- and the cpu_baseline leg of a one-rank run wants every core)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    affinity = (dp.pin_rank_to_gpu_numa(local_rank, local_world) if world > 1 and not args.no_pin
                else {"pinned": False, "how": "one rank: left to the OS" if world == 1 else "--no-pin"})
    dev = rt.device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.stub_engine:
            dist.init_process_group(rt.backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(rt.backend, rank=rank, world_size=world, device_id=dev)

    import synthetic
    if args.stub_engine:
        native = lib = None
    else:
        import native
        from decoding import WhisperDecoding
        from encoding import WhisperEncoding
        lib = native.load_library()

    # ---- engines: rank 0 builds once per box, everybody loads its own replica ----------------------
    eng_dir = Path(args.engine_cache) / f"{args.model}-{args.config}-seed{args.seed}"
    build_s = 0.0
    if rank == 0 and not args.stub_engine and not (eng_dir / "decoder_config.json").exists():
        eng_dir.parent.mkdir(parents=True, exist_ok=True)
        build_s = build_engines(args, eng_dir)
    if use_dist:
        dist.barrier()
    if args.stub_engine:
        enc = dec = _StubEngines()
    else:
        enc, dec = WhisperEncoding(eng_dir), WhisperDecoding(eng_dir)
    dec.sample_len = args.decode_steps
    if args.eager_loop:
        dec.use_graphs = False
    if args.groups > 0:
        dec.micro_batches = args.groups
    if os.environ.get("WM_GRAPH_PREFILL") == "0":       # A/B: the language pass and the prefill issued eagerly for every batch (rounds 1-5)
        dec.graph_prefill = False
    dims = synthetic.DIMS[args.model]
    B, T = args.batch, args.decode_steps

    # ---- inputs, resident before timing.  Default: every rank draws ITS OWN shard, N(0, 0.5) clipped to [-0.5, 1.5] like
    # synthetic.synthetic_mel, from (seed 1234, rank) on its own device -- utterances are independent, nothing has to travel (rank 0
    # keeps seed 1234 itself: a one-rank job draws what it always drew).  --scatter-inputs: the round-1..5 path, rank 0 draws the
    # global batch (2.2 GB at 8 x 576 clips) and scatters it over RCCL -- what a job that reads its audio on one rank would do.
    n_total = B * world
    feat = (dims["n_mels"], 2 * dims["n_audio_ctx"])

    def draw(n, seed):
        g = rt.generator(dev, seed)
        return (torch.randn((n, *feat), generator=g, device=dev) * 0.5).clamp_(-0.5, 1.5).half()
    t_in = time.perf_counter()
    if args.scatter_inputs:
        mels = draw(n_total, 1234) if rank == 0 else None
        mel = dp.scatter_utterances(mels, n_total, feat, torch.float16, dev).contiguous()
        del mels
    else:
        lo_, hi_ = dp.shard_bounds(n_total, rank, world)
        mel = draw(hi_ - lo_, dp.rank_seed(1234, rank))
    rt.sync()
    inputs = {"mode": "scatter from rank 0" if args.scatter_inputs else "every rank draws its own shard from (seed, rank)",
              "ms": round((time.perf_counter() - t_in) * 1e3, 1), "bytes_per_rank": int(mel.numel() * mel.element_size())}
    width = dec.initial_token_length + T

    first_token_events = []     # per step: (step start, encoder output ready, first sampled token exists, encoder was prefetched)
    loop_events = []            # (start, end) of every decode loop: torch events on the current stream, which main_loop
    last = {}                   # joins with its group streams before it returns
    stage_events = []           # per step: the stage boundaries (see step())
    enc.time_prefetch = True    # events around every prefetched encoder pass and around collect()'s wait (encoding.py)

    # One step = encoder + cross-K/V + language pass + prefill + decode loop of one batch.  Consecutive steps are software-
    # pipelined the way a transcription job over many batches is (run.py): while the HBM-bound decode loop of batch n runs,
    # the MFMA-bound encoder of batch n + 1 runs beside it on a budget of CUs (WhisperEncoding.prefetch).  Every step's
    # encoder runs inside the region that is timed: the first one in the open, the last decode loop with nothing beside it.
    def mark():
        e = rt.event()
        e.record()
        return e

    def step(more_to_come: bool):
        # stage boundaries as events on the current stream (main_loop and detect_language join their group streams into it
        # before they return): s0 | encoder in the open, or the wait for the prefetched one | s1 | cross-K/V projection +
        # language pass | s2 ... e0 | prefill + decode loop | e1 | gather | s3
        s0 = mark()
        was_prefetched = bool(last.get("prefetched"))
        xa = enc.collect() if was_prefetched else enc.get_audio_features_async(mel)
        s1 = mark()
        dec.detect_language(xa)
        s2 = mark()
        last["prefetched"] = bool(args.encoder_cus > 0 and more_to_come)
        if last["prefetched"]:
            enc.prefetch(mel, args.encoder_cus)
        e0 = mark()
        ft = rt.event()
        dec.first_token_event = ft
        tokens, sum_lp, _ = dec.main_loop(xa, ignore_eot=True)
        dec.first_token_event = None
        enc.loop_ended()            # the prefetched pass gives back its CU budget once the GPU is past this point
        e1 = mark()
        loop_events.append((e0, e1, last["prefetched"]))
        first_token_events.append((s0, s1, ft, was_prefetched))
        last["xa"] = xa
        res = dp.gather_results(tokens, sum_lp, n_total, width, dec.tokenizer.eot)
        stage_events.append((s0, s1, s2, e0, e1, mark(), was_prefetched))
        return res

    for k in range(args.warmup):
        out = step(k + 1 < args.warmup)
    if args.encoder_cus > 0 and hasattr(enc, "_side_stream") and getattr(enc, "_prefetch_stream", None) is None:
        enc._prefetch_stream = enc._side_stream(dev)      # (creating the helper's stream costs ~ 6 ms once per process: not inside the first timed step)
    if use_dist:                    # (the timed region runs the product's default schedule: no sampler, no group-by-group pass)
        dist.barrier()
    rt.sync()
    loop_events.clear()
    stage_events.clear()
    first_token_events.clear()
    enc.prefetch_events = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(k + 1 < args.steps)
    rt.sync()
    if use_dist:
        dist.barrier()
    my_elapsed = time.perf_counter() - t0
    elapsed = dp.max_over_ranks(my_elapsed, dev)
    per_rank_ms = dp.all_ranks(my_elapsed / args.steps * 1e3, dev)       # every rank's own ms per step: imbalance shows at a glance
    per_rank_cpus = dp.all_ranks(float(affinity.get("cpus") or 0), dev)
    per_rank_first = dp.all_ranks(float(affinity.get("first_cpu") if affinity.get("first_cpu") is not None else -1), dev)
    per_rank_input_ms = dp.all_ranks(inputs["ms"], dev)
    gathered = None
    if rank == 0 and out is not None:      # what the last step's gather brought to rank 0: rows of every rank, in utterance order
        g_tok, g_lp = out
        gathered = {"rows": int(g_tok.shape[0]), "width": int(g_tok.shape[1]),
                    "token_checksum": int(g_tok.to(torch.int64).sum().item() % (1 << 31)), "rows_expected": int(n_total)}

    # ---- where a step's time goes (GPU events of the timed steps; the review's "make the line tell the truth about the schedule") ----
    def _mean(xs):
        return round(float(np.mean(xs)), 2) if len(xs) else None

    def _span(a, b):             # ms between two events, None when one of them was never recorded (the stand-in decoder of --stub-engine)
        try:
            return a.elapsed_time(b)
        except Exception:        # noqa: BLE001
            return None
    pre = list(getattr(enc, "prefetch_events", []))
    pipeline = {
        "encoder_in_the_open_ms": _mean([a[0].elapsed_time(a[1]) for a in stage_events if not a[6]]),
        "encoder_prefetch_ms": _mean([t0e.elapsed_time(t1e) for t0e, t1e, _, _ in pre]),
        "collect_wait_ms": _mean([a[0].elapsed_time(a[1]) for a in stage_events if a[6]]),
        "cross_kv_and_language_ms": _mean([a[1].elapsed_time(a[2]) for a in stage_events]),
        "prefill_and_decode_loop_ms": _mean([a[3].elapsed_time(a[4]) for a in stage_events]),
        "gather_ms": _mean([a[4].elapsed_time(a[5]) for a in stage_events]),
        "other_ms": _mean([a[2].elapsed_time(a[3]) for a in stage_events]),
        "sum_per_step_ms": round(sum(a[0].elapsed_time(a[5]) for a in stage_events) / max(len(stage_events), 1), 2),
        # latency to the FIRST sampled token of a batch (W/run.py's user waits for this before any text exists): from the encoder's output
        # being ready (cross-K/V projection + language pass + 3-token prefill + the first greedy step) and from the start of a step whose
        # encoder ran in the open (the whole path from the mel)
        "first_token_after_encoder_ms": _mean([x for x in (_span(a[1], a[2]) for a in first_token_events) if x is not None]),
        "first_token_from_mel_ms": _mean([x for x in (_span(a[0], a[2]) for a in first_token_events if not a[3]) if x is not None]),
        "note": "means over the timed steps, GPU events on the stream that drives a step: the first step's encoder runs in the open, "
                "the others beside the previous step's decode loop (encoder_prefetch_ms = that pass on its own stream, start to end; "
                "collect_wait_ms = what the next step still waits for it after the loop has ended); sum_per_step_ms adds the stages of "
                "a step (encoder in the open or the wait, cross-K/V projection + language pass, prefill + decode loop, gather) and "
                "is to be read against ms_per_step (host clock over all steps)",
    }

    # ---- untimed probes after the headline region -------------------------------------------------------------------
    # (0) the dominant kernel's launch duration, HIP events stamped with the dispatch's own begin / end (wm_profile_*), on the
    # stream each launch runs on.  Inside the timed steps the launches are nodes of replayed graphs that three groups issue at
    # once: no event can sit on them (their in-situ duration is read from device-clock stamps further down).  So the same
    # launches are repeated right here, on the same buffers, eagerly and one group after the other -- which is also how
    # rocprofv3 --kernel-trace times them (it serialises the queues):
    #   in_loop   every cross-attention launch of 4 token steps behind the 3-token prefill (3 groups x 32 layers x 4 = 384):
    #             the launch behind its chain of short kernels, as the decode loop issues it            -> roofline.frac
    #   best      the launches of two language-ID passes, the groups taking turns: a launch behind another K/V launch,
    #             requests already streaming                                                             -> roofline.frac_best_case
    kernel_ms = {}
    if not args.no_roofline:
        def read_samples():
            ms, cnt = C.c_double(), C.c_int64()
            torch.cuda.synchronize()
            native.check(lib.wm_profile_read(C.byref(ms), C.byref(cnt), 1))
            return (ms.value / cnt.value if cnt.value else None), int(cnt.value)
        keep = (dec.use_graphs, dec.sample_len, dec.lang_id_sequential, dec.groups_sequential)
        try:
            # the probes step the groups one AFTER the other (a launch is timed with the chip to itself); groups that run side by side in the
            # timed region never take the one-launch step there, so they must not take it here either (16 utterances = 2 x 8 rows)
            dec.force_not_alone = dec._groups(B)[0] > 1
            native.check(lib.wm_profile_configure(1, 1, 8192))
            dec.lang_id_sequential = dec.profile_eager_passes = True      # (eager launches: the profiling events sit on them)
            for _ in range(2):
                dec.detect_language(last["xa"])
            kernel_ms["best"] = read_samples()
            dec.lang_id_sequential, dec.profile_eager_passes = keep[2], False
            dec.use_graphs, dec.groups_sequential, dec.sample_len = False, True, 5
            dec.main_loop(last["xa"], ignore_eot=True)
            kernel_ms["in_loop"] = read_samples()
        finally:
            dec.use_graphs, dec.sample_len, dec.lang_id_sequential, dec.groups_sequential = keep
            dec.force_not_alone = dec.profile_eager_passes = False
            native.check(lib.wm_profile_configure(0, 1, 0))
    # (a) the encoder alone on the whole chip (MFMA roofline of the other big stage)
    enc_alone_ms = None
    if not args.no_roofline:
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            enc.get_audio_features_async(mel)
            b.record()
            torch.cuda.synchronize()
            enc_alone_ms = a.elapsed_time(b)
    # (b) the SECOND figure: a LibriSpeech-sized job whose utterances end after LibriSpeech-like token counts (per-row
    # completion).  5 x B lengths are drawn, sorted (summarize.py batches clips by duration) and cut into 5 batches; inside a
    # batch the rows are dealt over the decoder's utterance groups (WhisperDecoding.balanced_order).  Every batch runs the
    # whole path like the headline's steps (one stage after the other); the first batch's pass is repeated once untimed
    # before the clock starts (it captures the graphs of the live-row loop).
    ragged = None
    if args.length_dist != "forced":
        n_jobs = 5
        all_limits = librispeech_like_lengths(n_jobs * B, T)
        deal = np.asarray(dec.balanced_order(B))
        batches = [all_limits[k * B:(k + 1) * B][deal] for k in range(n_jobs)]

        def ragged_step(limits):
            xa_r = enc.get_audio_features_async(mel)
            dec.detect_language(xa_r)
            e0, e1 = rt.event(), rt.event()
            e0.record()
            tokens, sum_lp, _ = dec.main_loop(xa_r, row_limit=torch.as_tensor(limits, dtype=torch.int32))
            e1.record()
            return tokens, e0, e1
        tokens_r, _, _ = ragged_step(batches[0])
        rt.sync()
        eot = dec.tokenizer.eot
        got = (tokens_r[:, dec.sample_begin:] != eot).sum(dim=1).cpu().numpy()
        if use_dist:
            dist.barrier()
        t_r = time.perf_counter()
        loops = []
        for limits in batches:
            _, e0, e1 = ragged_step(limits)
            loops.append((e0, e1))
        rt.sync()
        if use_dist:
            dist.barrier()
        r_elapsed = dp.max_over_ranks(time.perf_counter() - t_r, dev)
        loop_ms = [a.elapsed_time(b) for a, b in loops]
        # the same five batches PIPELINED as the headline's steps are (summarize.py --overlap_encoder): the encoder of batch n + 1 on
        # --encoder-cus CUs beside the decode loop of batch n
        rp_elapsed, released = None, []
        if args.encoder_cus > 0:
            if use_dist:
                dist.barrier()
            rt.sync()
            t_p = time.perf_counter()
            xa_p = enc.get_audio_features_async(mel)
            for i, limits in enumerate(batches):
                dec.detect_language(xa_p)
                if i + 1 < len(batches):
                    enc.prefetch(mel, args.encoder_cus)
                dec.main_loop(xa_p, row_limit=torch.as_tensor(limits, dtype=torch.int32))
                enc.loop_ended()
                if i + 1 < len(batches):
                    xa_p = enc.collect()
                    released.append(enc.last_release_layer)
            rt.sync()
            if use_dist:
                dist.barrier()
            rp_elapsed = dp.max_over_ranks(time.perf_counter() - t_p, dev)
        # audio seconds behind the token counts (the inverse of librispeech_like_lengths: 3.6 tokens/s + 2 timestamps)
        audio_s = float(np.maximum(all_limits - 2, 1).sum() / 3.6) * world
        best = min(r_elapsed, rp_elapsed) if rp_elapsed else r_elapsed
        kvb = 1 if args.config == "int8x" else 2
        row_bytes = dims["n_text_layer"] * dims["n_text_head"] * 2 * dims["n_audio_ctx"] * 64 * kvb       # cross K/V of one utterance, one token
        live_bytes = [float(l.sum() + B) * row_bytes for l in batches]     # a row is streamed once per token it samples (+ the step that ends it)
        ragged = {"length_dist": "librispeech-like: 5 x B lengths (log-normal durations, 3.6 tokens/s + 2 timestamps; bench.py: librispeech_like_lengths), "
                                 "sorted and cut into 5 batches as summarize.py batches clips by duration, rows dealt over the utterance groups",
                  "useful_tokens_per_s": round(float(all_limits.sum()) * world / r_elapsed, 1),
                  "tokens_per_utterance": {"mean": round(float(all_limits.mean()), 1), "min": int(all_limits.min()), "max": int(all_limits.max()),
                                           "per_batch_max": [int(l.max()) for l in batches]},
                  "tokens_match_the_limits": bool((got == batches[0]).all()),
                  "ms_per_batch": round(r_elapsed * 1e3 / n_jobs, 1), "decode_loop_ms": [round(x, 1) for x in loop_ms],
                  "ms_per_batch_pipelined": round(rp_elapsed * 1e3 / n_jobs, 1) if rp_elapsed else None,
                  "pipelined_encoder_released_at_layer": released or None,
                  "useful_tokens_per_s_pipelined": round(float(all_limits.sum()) * world / rp_elapsed, 1) if rp_elapsed else None,
                  "audio_seconds": round(audio_s, 1), "rtf": round(best / audio_s, 6),
                  "test_clean_estimate_s": round(best * 2620.0 / (n_jobs * B * world), 2),
                  "test_clean_note": "LibriSpeech test-clean is 2 620 utterances; the reference publishes 1 333 s for it on one A10 at batch 1 "
                                     "(BASELINE.md: TRT-LLM fp16, plugin flags; other hardware, fp16, real weights) -- this is the time of the "
                                     f"faster of the two schedules above scaled from {n_jobs * B * world} synthetic clips to 2 620 (context, not a same-node comparison)",
                  "decode_loop_vs_live_row_bytes": round(sum(loop_ms) * 1e-3 / (sum(live_bytes) / 5.66e12), 3),
                  "decode_loop_vs_live_row_bytes_per_batch": [round(m * 1e-3 / (b / 5.66e12), 3) for m, b in zip(loop_ms, live_bytes)],
                  "note": "whole path per batch as in the headline (encoder + cross-K/V + language pass + prefill + decode loop), one stage after "
                          "the other; rows that reach their length emit EOT, drop out of the attention kernels (live-row lists) and finished "
                          "groups are no longer stepped.  decode_loop_vs_live_row_bytes = decode loop time / (cross-K/V bytes of the rows still "
                          "decoding, summed over the steps, / 5.66 TB/s -- the rate of the forced loop's whole step)"}

    # decode loops with nothing beside them (the last step's; every step's when the pipelining is off) and with the next
    # batch's encoder beside them
    alone = [a.elapsed_time(b) for a, b, shared in loop_events if not shared]
    shared = [a.elapsed_time(b) for a, b, shared in loop_events if shared]
    decode_loop_ms = float(np.mean(alone)) if alone else None
    decode_loop_shared_ms = float(np.mean(shared)) if shared else None
    weight_bytes = sum(sess.engine.weight_bytes for sess in (enc.session, dec.decoder_session, dec.cross_attn_session))
    torch_bytes, device_in_use = rt.memory(dev)
    roofline = None
    if not args.no_roofline:
        avg_ms, n_samples = kernel_ms.get("in_loop") or (None, 0)
        best_ms, n_best = kernel_ms.get("best") or (None, 0)
        if avg_ms is None and best_ms is not None:        # (a loop of one token: only the language pass launched the kernel)
            avg_ms, n_samples = best_ms, n_best
        if avg_ms is not None:
            H, Tk = dims["n_text_head"], dims["n_audio_ctx"]
            n_micro, bounds = dec._groups(B)
            group = bounds[0][1] - bounds[0][0]                  # utterances per launch (stream-parallel groups)
            kv_bytes = 1 if args.config == "int8x" else 2
            algo_bytes = group * H * 2 * Tk * 64 * kv_bytes      # K and V of every (utterance, head), once (fp16; int8 in the opt-in mode)
            achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
            ref = rocprof_reference()
            roofline = {"kernel": "attn_cross_kernel (decode cross-attention)", "bound": "hbm",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "frac_best_case": round(algo_bytes / (best_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if best_ms else None,
                        # HBM bytes per launch: read from the PMC pass committed under profiles/ (FETCH_SIZE x 1024 x 2, the
                        # gfx950 correction for 16 B/lane streaming reads, MI355X_MICROARCH.md), per utterance-layer
                        **((not args.no_measure_traffic and world == 1 and measure_traffic(group, kv_bytes)) or
                           {k: (v if k != "traffic_source" or v is None else v + " (committed PMC pass, not measured in this run)")
                            for k, v in pmc_traffic(group, kv_bytes).items()}),
                        "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(avg_ms, 5),
                        "samples": n_samples, "best_case_launch_ms": round(best_ms, 5) if best_ms else None, "best_case_samples": n_best,
                        "utterances_per_launch": group,
                        **ref,
                        "rocprof_frac": (round(algo_bytes / (ref["rocprof_avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                         if ref.get("rocprof_avg_launch_ms") else None),
                        "rocprof_alone_frac": (round(algo_bytes / (ref["rocprof_alone_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                               if ref.get("rocprof_alone_launch_ms") and group == 192 else None),
                        "note": "achieved / frac: algorithmic bytes of a launch / its mean duration over EVERY cross-attention launch of 4 token "
                                "steps of all utterance groups (HIP events carrying the dispatch's own begin / end, on the launch's stream), "
                                "repeated right after the timed region on the same buffers with eager launches, one group after the other -- "
                                "inside the timed steps the launches are nodes of replayed graphs, which no event can sit on, and rocprofv3 "
                                "serialises the queues in the same way: rocprof_avg_launch_ms is the average over every launch of the kernel in "
                                "the committed rocprofv3 --kernel-trace --stats run of this round (bench.py --encoder-cus 0 --length-dist forced "
                                "--no-roofline: no launches beside the encoder, no ragged batches, no probes).  That mean is 5-6 % above the live figure "
                                "and the trace of the same command says why (rocprof_trace_source, scripts/trace_kernel_hist.py): the profiler does not "
                                "serialise the groups' queues completely -- rocprof_overlapped_share of the launches run beside another group's launch and "
                                "take twice as long; the launches ALONE on the chip average rocprof_alone_launch_ms, which is what the events measure "
                                "(rocprof_alone_frac against frac; rocprof_frac is the mean over everything).  frac_best_case: the "
                                "launches of two language-ID passes, groups taking turns (a launch that starts behind another K/V launch instead "
                                "of behind a chain of short kernels).  In the timed region itself "
                                f"{n_micro} groups replay the same kernel and grid at once and share the HBM: in_situ_*.  The launch is "
                                "persistent (<= 2 workgroups per CU, every workgroup the same number of items) and software-"
                                "pipelined: with 8 of a CU's 32 wave slots it leaves room for the other groups' short kernels"}
            # ---- the same kernel IN SITU: graph-replayed launches, the groups sharing the HBM (device-side stamps around
            # every launch, wm_debug_timeline; measured on extra, untimed decode steps after the timed region) ----
            roofline.update(in_situ_probe(dec, lib, last["xa"], B, n_micro, algo_bytes))
            if args.encoder_cus > 0 and args.steps > 1:
                b = in_situ_probe(dec, lib, last["xa"], B, n_micro, algo_bytes, beside=(enc, mel, args.encoder_cus))
                roofline["in_situ_beside_encoder"] = {k.replace("in_situ_", ""): v for k, v in b.items() if k != "in_situ_note"}
        else:
            # utterance groups of up to eight rows: the cross-attention is a stage of the one-launch token step (gemv_chain.hip), no launch of the
            # K/V kernel exists to sample.  The token step as a whole is the unit then (decode_step_* below): achieved = its bytes / its time
            chain_traffic, chain_src = None, None        # HBM bytes per launch from the newest committed PMC pass of the same command (by round name)
            try:
                import glob as _glob
                import re as _re
                pmc = sorted(_glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_b{B}_chain_fetch.txt")), key=profile_round_key)
                txt = Path(pmc[-1]).read_text() if pmc else ""
                m = _re.search(r"streaming reads\): ([0-9.]+) MB", txt)
                if m and args.model == "large-v2" and args.config == "int8":
                    chain_traffic = int(float(m.group(1)) * 1e6)
                    chain_src = (f"{os.path.relpath(pmc[-1], ROOT)} (committed rocprofv3 --pmc FETCH_SIZE pass of bench.py --batch {B}, "
                                 "not measured in this run; FETCH_SIZE KB x 1024 x 2)")
            except OSError:
                pass
            roofline = {"kernel": "gemv_chain_kernel (the one-launch decode step of a group of 1-8 utterances: every layer's self-attention, Linears, cross-attention pieces and merge as stages of ONE launch)",
                        "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": chain_traffic, "traffic_source": chain_src,
                        "note": "groups of 1-8 utterances: achieved / frac are the whole token step's (decode_step_bytes.total over decode_step_ms), "
                                "a latency-bound chain of dependent stages, not a streaming kernel"}
        if roofline is not None:
            H, Tk = dims["n_text_head"], dims["n_audio_ctx"]
            n_micro, bounds = dec._groups(B)
            kv_bytes = 1 if args.config == "int8x" else 2
            if decode_loop_ms is not None:
                step_ms = decode_loop_ms / T
                cross_bytes = B * dims["n_text_layer"] * H * 2 * Tk * 64 * kv_bytes          # B x 245.76 MB at large-v2
                # ALL of SURVEY 8d's bytes of one token step: the decoder's Linear weights and the logits matrix as resident (every
                # utterance group streams them once per step), the cross K/V of every utterance, the self-attention cache at the
                # loop's mean length (prefill + T / 2 tokens; one byte per element with the int8 cache, two without)
                wo_cfg, i8kv_cfg = CONFIGS[args.config]
                Ct, Lt, V = dims["n_text_state"], dims["n_text_layer"], dims["n_vocab"]
                lin_bytes = Lt * 14 * Ct * Ct * (0.5 if wo_cfg == "int4" else 1 if wo_cfg else 2)      # per layer: qkv 3 + out 1, cross q 1 + out 1, mlp 8 (x C^2; the cross k / v projections live in their own engine): 734.0 MB int8 at large-v2
                logit_bytes = V * Ct * 2
                self_bytes = B * Lt * 2 * Ct * (dec.initial_token_length + T / 2.0) * (1 if i8kv_cfg else 2)
                # SURVEY 8d counts W_dec + W_logits ONCE per step whatever the schedule; the stream-parallel groups each stream them again --
                # bytes that move but are not useful work, so they are NOT in the fraction (rounds 1-5 multiplied the
                # weights by the group count and so credited the re-reads); the re-read factor and the as-streamed total stand beside it
                all_bytes = (lin_bytes + logit_bytes) + cross_bytes + self_bytes
                streamed_bytes = n_micro * (lin_bytes + logit_bytes) + cross_bytes + self_bytes
                roofline.update({"decode_loop_ms": round(decode_loop_ms, 2), "decode_step_ms": round(step_ms, 3),
                                 "decode_step_frac": round(all_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "decode_step_bytes": {"linear_weights": int(lin_bytes), "logits_matrix": int(logit_bytes),
                                                       "cross_kv": int(cross_bytes), "self_kv_mean": int(self_bytes), "total": int(all_bytes),
                                                       "utterance_groups": n_micro, "weight_reread_factor": n_micro,
                                                       "total_as_streamed": int(streamed_bytes)},
                                 "decode_step_frac_as_streamed": round(streamed_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "decode_step_frac_cross_kv_only": round(cross_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "decode_step_note": "SURVEY 8d bytes of one token step -- Linear weights + logits matrix ONCE + cross K/V of the "
                                                     "whole batch + self-attention cache at the loop's mean length -- / (decode loop time / tokens) / 8 TB/s; "
                                                     "every utterance group streams the weights again (weight_reread_factor): those bytes are in "
                                                     "total_as_streamed / decode_step_frac_as_streamed only"
                                                     + ("; decode loops with nothing beside them (the last step's)" if shared else "")})
                if roofline["achieved"] is None:
                    roofline["achieved"] = round(all_bytes / (step_ms * 1e-3) / 1e9, 1)
                    roofline["frac"] = roofline["decode_step_frac"]
                if decode_loop_shared_ms is not None:
                    roofline.update({"decode_loop_beside_encoder_ms": round(decode_loop_shared_ms, 2),
                                     "decode_step_beside_encoder_ms": round(decode_loop_shared_ms / T, 3)})
                # the whole job against its HBM-only floor: every token step streams the batch's cross K/V once
                ms_step = elapsed / args.steps * 1e3
                roofline["job_frac"] = round(cross_bytes * T / (HBM_PEAK_GBS * 1e9) / (ms_step * 1e-3), 4)
                roofline["job_note"] = "cross-K/V bytes of all token steps of one batch / 8 TB/s, over the measured time per step (encoder, projection, language pass and prefill included in the time, not in the bytes)"
            if enc_alone_ms is not None:
                fl = encoder_flops_per_clip(dims) * B
                roofline["encoder"] = {"bound": "mfma", "ms": round(enc_alone_ms, 1), "achieved": round(fl / (enc_alone_ms * 1e-3) / 1e12, 1),
                                       "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / (enc_alone_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                       "note": f"one encoder pass of {B} clips alone on the chip after the timed region (HIP events): "
                                               f"{encoder_flops_per_clip(dims) / 1e12:.3f} TFLOP per clip (SURVEY 8d) against the dense fp16 MFMA peak"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * T / (ms_per_step * 1e-3)
        result = {
            "metric": "decode tokens/s, whole job (encoder + cross-KV + language-ID + prefill + greedy decode), "
                      f"Whisper {args.model} " + ("int8 weight-only + int8 KV" if args.config == "int8" else args.config)
                      + "; rtf reported beside it",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2),
            "ms_per_step_per_rank": {"min": round(min(per_rank_ms), 2), "max": round(max(per_rank_ms), 2), "all": [round(x, 2) for x in per_rank_ms]},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "rtf": round((ms_per_step * 1e-3) / (B * 30.0), 6),
            "utterances_per_s": round(n_total / (ms_per_step * 1e-3), 2),
            "config": {"workload": f"whisper {args.model} {args.config}: {WORKLOADS[args.config]}; {B} x 30 s utterances per GPU per step, {T} forced greedy tokens each "
                                   f"(+3-token prefill, +1-token language-ID pass); random-init weights",
                       "batch_per_gpu": B, "decode_steps": T,
                       "pipelining": (f"encoder of step n+1 on {args.encoder_cus} CUs beside the decode loop of step n; all {args.steps} encoders inside the timed region"
                                      if args.encoder_cus > 0 and args.steps > 1 else "none"),
                       "parallelism": f"dp{world} (utterance sharding, no "
                                                                            f"data-path collective)"},
            "engine_build_s": round(build_s, 1),
            # multi-rank hygiene: how the inputs reached the ranks (outside the timed region), where each rank's process sits on the host,
            # what the last gather delivered, and whether a one-launch decode step was declined or gave up during this run
            "inputs": dict(inputs, ms_per_rank=[round(x, 1) for x in per_rank_input_ms]),
            "affinity": dict(affinity, cpus_per_rank=[int(x) for x in per_rank_cpus], first_cpu_per_rank=[int(x) for x in per_rank_first]),
            "gathered": gathered,
            "decode_chain": (None if native is None else {k: v for k, v in native.chain_status().items() if k != "footprint"}),
            # what sits in HBM while the job runs: engine weights (hipMalloc'ed by the library: weight-only encoder / cross-K/V
            # matrices are resident ONCE, as their fp16 expansion), everything torch allocated for the path (mel, encoder output,
            # KV cache, cross K/V, logits, workspaces), and the device-level figure the reference's memory chart reports
            # (README.md:178-180: 9.3-11.3 GB at batch 1 on an A10)
            "hbm_bytes_resident": {"engine_weights": int(weight_bytes), "buffers": int(torch_bytes),
                                   "device_in_use": int(device_in_use), "batch_per_gpu": B},
            "pipeline": pipeline,
            "roofline": roofline,
            "second_figure": ragged,
            # arithmetic-order / scheduling knobs read from the environment, echoed when set (defaults otherwise)
            "env_knobs": {k: v for k, v in sorted(os.environ.items()) if k.startswith("WM_")} or None,
            # what the package did to the HIP runtime's environment (and whether in time), the lab knobs the LIBRARY honoured (WM_LAB=1 only)
            "runtime": native.runtime_report() if native is not None else None,
            "hip_runtime_knobs": {k: os.environ[k] for k in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "GPU_MAX_HW_QUEUES", "AMD_OPT_FLUSH") if k in os.environ},
        }
    # every rank is done with the GPU and with its peers BEFORE rank 0 starts anything long on the host: the last barrier and the process
    # group's teardown come first, the cpu_baseline leg (one rank only: ~ 150 s of host work) and the line afterwards -- no rank ever sits
    # in a collective waiting for rank 0's CPU work
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            result.update(wer_block(args))          # "not measured" unless real weights and LibriSpeech are on the box
            result["cpu_baseline"] = cpu_baseline(args, T)
        if json_fd is None:
            print(json.dumps(result), flush=True)
        else:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
