"""Host time against wall time of replaying a captured decode-step graph (B = 1 by default): is the replay of ~300 kernel
nodes issue-bound on the host when the runtime enqueues the nodes one by one (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import native  # noqa: F401
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
dec.sample_len = 128
for B in (int(x) for x in (sys.argv[1:] or ["1"])):
    mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
    xa = enc.get_audio_features_async(mel)
    dec.detect_language(xa)
    dec.main_loop(xa, ignore_eot=True); dec.main_loop(xa, ignore_eot=True)
    st = dec._state[B]
    for gkey, graph in st['graphs'].items():
        torch.cuda.synchronize()
        s = dec._group_streams(1, xa.device)[0] if hasattr(dec, "_group_streams") else torch.cuda.current_stream()
        with torch.cuda.stream(s):
            st['counters'][gkey].fill_(5)
            for n in (1, 20, 120):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                st['counters'][gkey].fill_(3)
                for _ in range(n): graph.replay()
                ti = time.perf_counter() - t0
                torch.cuda.synchronize(); t1 = time.perf_counter() - t0
                print(f"B={B} graph {gkey}: {n} replays: host {ti / n * 1e3:.3f} ms each, wall {t1 / n * 1e3:.3f} ms each  (capture={os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')})", flush=True)
            st['counters'][gkey].fill_(5)
        break
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a.record(); dec.main_loop(xa, ignore_eot=True); b.record()
        ti = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"B={B} main_loop of {dec.sample_len} tokens: host {ti * 1e3:.1f} ms, events {a.elapsed_time(b):.1f} ms = {a.elapsed_time(b) / dec.sample_len:.3f} ms per token", flush=True)
