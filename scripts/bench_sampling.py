"""Round 4: temperature / best_of through the fused device loop.  best_of = 5 candidates of 64 utterances (320 rows, the draw inside
the greedy kernel) against the greedy loop over 320 utterances (the same rows, arg-max), and against the literal host loop the
options used to take (main_loop_reference: one decode() per token, host filters, torch's Categorical).  Engines: the ones bench.py
caches (large-v2, weight-only int8 + int8 KV).      python scripts/bench_sampling.py [tokens=32]"""
import os, sys, time
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import native  # noqa: F401
import torch
import synthetic
from decoding import DecodingOptions, WhisperDecoding
from encoding import WhisperEncoding
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
assert (eng / "decoder_config.json").exists(), "run bench.py once first: it builds and caches the engines"
dims = synthetic.DIMS["large-v2"]
enc = WhisperEncoding(eng)
g = torch.Generator(device="cuda").manual_seed(1234)
mel = (torch.randn((320, dims["n_mels"], 2 * dims["n_audio_ctx"]), generator=g, device="cuda") * 0.5).clamp_(-0.5, 1.5).half()
xa = enc.get_audio_features_async(mel)
torch.cuda.synchronize()


def timed(dec, feats, reps=3, **kw):
    dec.detect_language(feats)
    dec.main_loop(feats, **kw)                    # warm-up: graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dec.main_loop(feats, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


greedy = WhisperDecoding(eng, options=DecodingOptions(sample_len=T))
ms_g = timed(greedy, xa, ignore_eot=True)
del greedy
torch.cuda.empty_cache()
samp = WhisperDecoding(eng, options=DecodingOptions(temperature=0.7, best_of=5, sample_len=T))
xa64 = xa[:64].contiguous()
ms_s = timed(samp, xa64, ignore_eot=True)
samp.device_sampling = False
samp.sample_len = min(T, 8)
t0 = time.perf_counter()
samp.main_loop(xa64)
torch.cuda.synchronize()
ms_h = (time.perf_counter() - t0) * 1e3 / samp.sample_len * T
print(f"{T} tokens: greedy loop, 320 utterances {ms_g:.1f} ms ({ms_g / T:.2f} ms per token) | best_of 5 x 64 utterances, device draw {ms_s:.1f} ms "
      f"({ms_s / T:.2f} ms per token, {ms_s / ms_g:.2f} x greedy) | literal host loop {ms_h:.0f} ms ({ms_h / T:.1f} ms per token, from {samp.sample_len} tokens)")
