"""Diagnostic: decode-loop time vs number of stream-parallel micro-batches, and the host's issue time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, synthetic
from pathlib import Path
from decoding import WhisperDecoding
from encoding import WhisperEncoding
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
if not (eng / "decoder_config.json").exists():
    import argparse, bench
    eng.parent.mkdir(parents=True, exist_ok=True)
    bench.build_engines(argparse.Namespace(model="large-v2", config="int8", seed=0), eng)
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.sample_len = 32
d = synthetic.DIMS["large-v2"]
for B in (int(x) for x in sys.argv[1:] or ["64"]):
    mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
    xa = enc.get_audio_features_async(mel)
    dec.detect_language(xa)
    for nm in (1, 2, 4):
        dec.micro_batches = nm
        dec.main_loop(xa, ignore_eot=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = dec.main_loop(xa, ignore_eot=True)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(); t1 = time.perf_counter() - t0
        print(f"B={B} micro={nm}: loop {t1*1e3:.1f} ms ({t1/32*1e3:.3f} ms/step, {B*32/t1:.0f} tok/s), host issue {t_issue*1e3:.1f} ms", flush=True)
