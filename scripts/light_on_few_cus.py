"""Diagnostic (run under rocprofv3 --kernel-trace --stats): one utterance group's decode steps on a CU-masked
stream, to see which of the short kernels depend on having the whole chip.  usage: light_on_few_cus.py [n_cus=64] [B=128]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native, synthetic
from pathlib import Path
from decoding import WhisperDecoding
from encoding import WhisperEncoding
n_cus = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
if not (eng / "decoder_config.json").exists():
    import argparse, bench
    eng.parent.mkdir(parents=True, exist_ok=True)
    bench.build_engines(argparse.Namespace(model="large-v2", config="int8", seed=0), eng)
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
xa = enc.get_audio_features_async(mel)
dec.sample_len = 4
dec.micro_batches = 1
dec.detect_language(xa)
dec.main_loop(xa, ignore_eot=True)
st = dec._state[B]
stream = native.create_masked_stream([i < n_cus for i in range(256)], 0) if n_cus < 256 else torch.cuda.Stream()
cap = dec.decoder_config['num_text_ctx']
counter = torch.full((1,), 40, dtype=torch.int32, device="cuda")
io = dec.decoder_session.make_decoder_io(st['tokens'], dec.positional_embedding, st['cross'], st['kv'], cap, st['kv'], cap,
                                         st['logits'], 1, slot=7, n_past_dev=counter, n_new=1)
lib, h = dec.decoder_session._engine.lib, dec.decoder_session._engine.handle
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(5):
        native.check(lib.wm_decoder_step(h, C.byref(io), stream.cuda_stream))
    stream.synchronize()
    print(f"n_cus={n_cus} B={B}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per step (all kernels on the masked stream)", flush=True)
