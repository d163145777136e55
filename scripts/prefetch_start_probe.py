"""Round 6 probe: what the START of a prefetched encoder pass costs the caller (bench.py `pipeline.other_ms`: 2-6 ms at one clip).
Host clock around WhisperEncoding.prefetch(), and from its return to the first decoder call main_loop issues; GPU events around the same.
    python scripts/prefetch_start_probe.py [batch] [switch interval s]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import native  # noqa
import torch
from pathlib import Path
from decoding import WhisperDecoding
from encoding import WhisperEncoding
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if len(sys.argv) > 2:
    sys.setswitchinterval(float(sys.argv[2]))
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.sample_len = 32
g = torch.Generator(device="cuda").manual_seed(1234)
mel = (torch.randn((B, 80, 3000), generator=g, device="cuda") * 0.5).clamp_(-0.5, 1.5).half()
xa = enc.get_audio_features_async(mel)
dec.detect_language(xa); dec.main_loop(xa, ignore_eot=True); torch.cuda.synchronize()
stamps = {}
orig = dec.decoder_session.decoder_step
def spy(*a, **k):
    stamps.setdefault("first_step", time.perf_counter())
    return orig(*a, **k)
dec.decoder_session.decoder_step = spy
rows = []
for it in range(6):
    dec.detect_language(xa)
    torch.cuda.synchronize()
    stamps.clear()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    enc.prefetch(mel, 96)
    t1 = time.perf_counter()
    e1.record()
    t2 = time.perf_counter()
    dec.main_loop(xa, ignore_eot=True)
    t3 = time.perf_counter()
    enc.loop_ended(); enc.collect(); torch.cuda.synchronize()
    rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (stamps["first_step"] - t2), e0.elapsed_time(e1), 1e3 * (t3 - t2)))
print(f"B = {B}, switch interval {sys.getswitchinterval()} s: per iteration (ms): prefetch() call | event record | return -> first decoder_step issued | GPU span between the two events | main_loop host time")
for r in rows:
    print("   " + "  ".join(f"{x:8.3f}" for x in r))
