"""Experiment: the encoder of the NEXT batch on a budget of CUs (persistent GEMM workgroups that never leave their CU, so
the decode kernels are never dispatched behind them) while the decode loop of the current batch runs on the rest.

    python scripts/overlap_enc_dec2.py [batch] [tokens]
"""
import os, sys, time, threading
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 288
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
enc = WhisperEncoding(eng)
dec = WhisperDecoding(eng)
dec.sample_len = T
mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
se = torch.cuda.Stream()
sb = torch.cuda.Stream()
out = {}
def stage_a():
    with torch.cuda.stream(se):
        out["xa2"] = enc.get_audio_features_async(mel)
        se.synchronize()
xa = enc.get_audio_features(mel)
dec.detect_language(xa)
def stage_b():
    with torch.cuda.stream(sb):
        dec.main_loop(xa, ignore_eot=True)
        sb.synchronize()
stage_a(); stage_b(); torch.cuda.synchronize()
def wall(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
os.environ.pop("WM_GEMM_WGS", None)
ta = wall(stage_a); tb = wall(stage_b)
print(f"B={B} T={T}: encoder alone (full chip) {ta:.1f} ms, decode loop alone {tb:.1f} ms, sum {ta + tb:.1f}", flush=True)
def both():
    th = threading.Thread(target=stage_a)
    th.start()
    t0 = time.perf_counter()
    stage_b()
    out["tb"] = (time.perf_counter() - t0) * 1e3
    th.join()
for wgs in (256, 128, 96, 64, 32):
    os.environ["WM_GEMM_WGS"] = str(wgs)
    ta_m = wall(stage_a)
    both(); torch.cuda.synchronize()
    t = wall(both)
    print(f"  GEMM on {wgs} workgroups: encoder alone {ta_m:.1f} ms; together {t:.1f} ms (decode loop inside: {out['tb']:.1f} ms; one after the other on the full chip: {ta + tb:.1f})", flush=True)
