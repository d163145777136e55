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
for B in (int(x) for x in (sys.argv[1:] or ["16", "128"])):
    mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
    xa = enc.get_audio_features_async(mel)
    dec.detect_language(xa)
    for nm in (2, 3, 4):
        for g in (True,):
            dec.micro_batches, dec.use_graphs = nm, g
            dec.main_loop(xa, ignore_eot=True); dec.main_loop(xa, ignore_eot=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dec.main_loop(xa, ignore_eot=True)
            ti = time.perf_counter() - t0
            torch.cuda.synchronize(); t1 = time.perf_counter() - t0
            print(f"B={B} micro={nm} graphs={g}: {t1/32*1e3:.3f} ms/step, host issue {ti/32*1e3:.3f} ms/step", flush=True)
st = dec._state.get(128, {"graphs": {}})
gkey = (2, 0)
if gkey in st['graphs']:
    torch.cuda.synchronize()
    s = dec._streams[0]
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        st['counters'][gkey].fill_(5)
        for _ in range(10): st['graphs'][gkey].replay()
    ti = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print(f"10 replays of one group graph (B=64): host {ti*100:.3f} ms each, wall {t1*100:.3f} ms each")
