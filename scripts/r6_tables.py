"""Prints the two tables DESIGN.md section 5 / section 6 carry, from the round's final artefacts under profiles/ (r6z_*):
    python scripts/r6_tables.py ladder | numbers"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(name):
    return json.load(open(os.path.join(P, name)))


FORMS = {**{b: "one launch per step" for b in range(1, 9)}, 12: "fused small-batch", 16: "fused small-batch", 24: "fused small-batch", 32: "fused small-batch",
         64: "split-K chain", 128: "row-split", 256: "row-split"}


def ladder():
    print("| utterances | groups × rows | form | ms per token step | `decode_step_frac` (as streamed) | tokens/s whole job | first token after the encoder, ms | HBM traffic ÷ algorithmic bytes |")
    print("|---|---|---|---|---|---|---|---|")
    for b in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 64, 128, 256):
        d = load(f"r6z_bench_b{b}.json")
        r = d["roofline"]
        sb = r["decode_step_bytes"]
        g = sb["utterance_groups"]
        tr = r.get("traffic")
        if b <= 8 and tr:
            ratio = f"{tr / (sb['total'] - sb['logits_matrix']):.3f} (the launch = the step without the logits)"
        elif tr and r.get("algorithmic_bytes_per_launch"):
            ratio = f"{tr / r['algorithmic_bytes_per_launch']:.4f} (K/V launch)"
        else:
            ratio = "--"
        print(f"| {b} | {g} × {b // g} | {FORMS[b]} | {r['decode_step_ms']} | {r['decode_step_frac']:.3f} ({r['decode_step_frac_as_streamed']:.3f}) | {d['value']:.0f} | "
              f"{d['pipeline'].get('first_token_after_encoder_ms')} | {ratio} |")
    d = load("r6z_bench_line.json")
    r = d["roofline"]
    print(f"| 576 | 3 × 192 | row-split | {r['decode_step_ms']} | {r['decode_step_frac']:.3f} ({r['decode_step_frac_as_streamed']:.3f}) | {d['value']:.0f} | "
          f"{d['pipeline'].get('first_token_after_encoder_ms')} | {r['traffic'] / r['algorithmic_bytes_per_launch']:.4f} (K/V launch, measured in the run) |")


def numbers():
    d = load("r6z_bench_line.json")
    r, p, s, c = d["roofline"], d["pipeline"], d["second_figure"], d["cpu_baseline"]
    f = load("r6z_bench_force_dist.json")
    old = load("r5z_bench_line.json")
    ro, so = old["roofline"], old["second_figure"]
    rows = [
        ("headline (tokens/s, ms per step of 576 × 128 tokens)", f"{old['value']:.0f}, {old['ms_per_step']:.1f} (driver: `BENCH_r05.json` 16 669)", f"**{d['value']:.0f}**, {d['ms_per_step']:.1f}; `--force-dist` (RCCL on one rank): {f['value']:.0f}"),
        ("`roofline.frac` (live HIP events over every in-loop launch) / launch", f"{ro['frac']} / {ro['avg_launch_ms'] * 1e3:.1f} µs", f"**{r['frac']}** / {r['avg_launch_ms'] * 1e3:.1f} µs; best case {r['frac_best_case']}"),
        ("rocprofv3 mean of the dominant kernel, clean command (all launches / launches alone)", f"{ro.get('rocprof_avg_launch_ms', 0) * 1e3:.1f} / {ro.get('rocprof_alone_launch_ms', 0) * 1e3:.1f} µs",
         f"{(r.get('rocprof_avg_launch_ms') or 0) * 1e3:.1f} µs = frac {r.get('rocprof_frac')} / {(r.get('rocprof_alone_launch_ms') or 0) * 1e3:.1f} µs = {r.get('rocprof_alone_frac')} (`{r.get('rocprof_source')}`)"),
        ("`roofline.traffic` (FETCH_SIZE × 1024 × 2, measured in the run) ÷ algorithmic bytes", f"{ro['traffic'] / ro['algorithmic_bytes_per_launch']:.4f}", f"{r['traffic'] / r['algorithmic_bytes_per_launch']:.4f}"),
        ("token step alone / beside the encoder; `decode_step_frac`", f"{ro['decode_step_ms']} / {ro.get('decode_step_beside_encoder_ms')} ms; {ro['decode_step_frac']} (weights × 3 groups)",
         f"{r['decode_step_ms']} / {r.get('decode_step_beside_encoder_ms')} ms; **{r['decode_step_frac']}** on SURVEY 8d's bytes (weights once), {r['decode_step_frac_as_streamed']} as streamed"),
        ("encoder, 576 clips alone", f"{ro['encoder']['ms']:.0f} ms ({ro['encoder']['achieved']:.0f} TFLOP/s)", f"{r['encoder']['ms']:.0f} ms ({r['encoder']['achieved']:.0f} TFLOP/s, {r['encoder']['frac']} of the dense fp16 peak)"),
        ("cross-K/V projection + language pass", f"{old['pipeline']['cross_kv_and_language_ms']} ms", f"{p['cross_kv_and_language_ms']} ms"),
        ("first sampled token: after the encoder's output / from the mel", "--", f"{p['first_token_after_encoder_ms']} / {p['first_token_from_mel_ms']} ms (576 utterances; batch 1: see the ladder)"),
        ("second figure (LibriSpeech-like lengths): ms per batch sequential / pipelined, useful tokens/s", f"{so['ms_per_batch']} / {so['ms_per_batch_pipelined']}, {so['useful_tokens_per_s']:.0f}",
         f"{s['ms_per_batch']} / {s['ms_per_batch_pipelined']}, {s['useful_tokens_per_s']:.0f}; test-clean estimate {s['test_clean_estimate_s']} s on one GPU"),
        ("`cpu_baseline` (the oracle on the host, one full-depth large-v2 clip)", f"{old['cpu_baseline']['value']} tokens/s at {old['cpu_baseline']['cores']} threads",
         f"{c['value']} tokens/s at {c['cores']} threads (sweep in the run); tiny.en {c['tiny_en']['value']} tokens/s"),
        ("`wer`", "--", f"{d.get('wer')}"),
    ]
    print("| | round 5 (`profiles/r5z_bench_line.json`) | round 6 (`profiles/r6z_bench_line.json`) |")
    print("|---|---|---|")
    for a, b, c_ in rows:
        print(f"| {a} | {b} | {c_} |")


if __name__ == "__main__":
    (ladder if (sys.argv[1:] or ["ladder"])[0] == "ladder" else numbers)()
