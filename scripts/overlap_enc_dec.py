"""Experiment: can the encoder stage of the next batch hide under the decode loop of the current one?"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
enc = WhisperEncoding(eng)
decs = [WhisperDecoding(eng), WhisperDecoding(eng)]
for d in decs: d.sample_len = T
mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
xa = [None, None]
def stage_a(i, stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        xa[i] = enc.get_audio_features_async(mel)
        decs[i].detect_language(xa[i])
        torch.cuda.current_stream().synchronize()
def stage_b(i):
    decs[i].main_loop(xa[i], ignore_eot=True)
for i in (0, 1):
    stage_a(i); stage_b(i)                       # warm up both (graphs captured)
torch.cuda.synchronize()
def wall(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
ta = wall(lambda: stage_a(0)); tb = wall(lambda: stage_b(1))
print(f"B={B} T={T}: stage A (encoder + cross-KV + lang-id) alone {ta:.1f} ms, stage B (decode loop) alone {tb:.1f} ms, sum {ta+tb:.1f}")
import native
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
masks = {"all CUs": [True] * n_cu,
         "encoder on CUs 0..127": [i < 128 for i in range(n_cu)], "encoder on CUs 0..159": [i < 160 for i in range(n_cu)],
         "encoder on 16 of every 32": [(i % 32) < 16 for i in range(n_cu)], "encoder on 24 of every 32": [(i % 32) < 24 for i in range(n_cu)]}
for k, (name, mask) in enumerate(masks.items()):
    se = native.create_masked_stream(mask, 5 + k)               # a queue of its own, confined to the mask
    stage_a(0, se)                                             # warm
    torch.cuda.synchronize()
    ta_m = wall(lambda: stage_a(0, se))
    sb = torch.cuda.Stream()          # NOT the legacy default stream: it synchronises implicitly with every blocking stream (the masked ones are)
    def both():
        th = threading.Thread(target=stage_a, args=(0, se))
        th.start()
        with torch.cuda.stream(sb):
            stage_b(1)
        th.join()
    both(); torch.cuda.synchronize()
    t = wall(both)
    print(f"  {name}: stage A alone on that stream {ta_m:.1f} ms; A and B together {t:.1f} ms (A alone + B alone = {ta_m + tb:.1f}, full-chip sum {ta + tb:.1f})", flush=True)
