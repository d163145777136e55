"""Diagnostic: does the decode step time depend on the phase offset between the two utterance groups?
(hypothesis: in phase, both groups stream cross K/V together and then both sit in latency-bound chains)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch
from pathlib import Path
from decoding import WhisperDecoding
from encoding import WhisperEncoding
import synthetic
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
if not (eng / "decoder_config.json").exists():
    import argparse, bench
    eng.parent.mkdir(parents=True, exist_ok=True)
    bench.build_engines(argparse.Namespace(model="large-v2", config="int8", seed=0), eng)
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
STEPS = 64
dec.sample_len = STEPS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# calibrate torch.cuda._sleep
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
print(f"_sleep: {cyc_per_us:.1f} cycles/us", flush=True)
mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
xa = enc.get_audio_features_async(mel)
dec.detect_language(xa)
dec.micro_batches = 2
dec.main_loop(xa, ignore_eot=True)
for off_us in (0, 40, 80, 120, 160, 200, 300, 2000, 7000):
    dec._phase_sleep_cycles = int(off_us * cyc_per_us)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.main_loop(xa, ignore_eot=True)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"B={B} offset {off_us:5d} us: {best/STEPS*1e3:.3f} ms/step, {B*STEPS/best:.0f} tok/s", flush=True)
