"""Round 6: the one-launch step dispatched right at the START of an encoder pass on a side stream (no helper thread, no layer-by-layer issue):
which budget of the pass makes it give up?    python scripts/chain_beside_encoder_stress.py [loops] [cu_budget] [host delay ms before the loop]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import ctypes as C
import native  # noqa
import numpy as np
import torch
from pathlib import Path
import bench
from decoding import WhisperDecoding
from encoding import WhisperEncoding
LOOPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CB = int(sys.argv[2]) if len(sys.argv) > 2 else 96
DELAY = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
B = int(os.environ.get("B", "5"))
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
lib = native.load_library()
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.sample_len = 128
g = torch.Generator(device="cuda").manual_seed(1234)
mel = (torch.randn((B, 80, 3000), generator=g, device="cuda") * 0.5).clamp_(-0.5, 1.5).half()
limits_all = bench.librispeech_like_lengths(5 * B, 128)
deal = np.asarray(dec.balanced_order(B))
batches = [limits_all[k * B:(k + 1) * B][deal] for k in range(5)]
xa = enc.get_audio_features_async(mel)
dec.detect_language(xa); dec.main_loop(xa, row_limit=torch.as_tensor(batches[0], dtype=torch.int32)); torch.cuda.synchronize()
side = torch.cuda.Stream()
out = torch.empty_like(xa)
gave_up = 0
t0 = time.perf_counter()
for it in range(LOOPS):
    limits = batches[it % 5]
    dec.detect_language(xa)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        enc.get_audio_features_async(mel, out=out, cu_budget=CB)
    if DELAY:
        time.sleep(DELAY * 1e-3)
    dec.main_loop(xa, row_limit=torch.as_tensor(limits, dtype=torch.int32))
    torch.cuda.synchronize()
    st = native.chain_status()
    if st["declined"] or st["error_pending"]:
        gave_up += 1
        print(f"  loop {it}: GAVE UP", flush=True)
        err = C.c_int(0); lib.wm_decode_chain_error(C.byref(err)); lib.wm_set_decode_chain(-1)
print(f"B={B} encoder pass on a side stream, cu_budget {CB}, host delay {DELAY} ms: {gave_up} give-ups in {LOOPS} loops, {(time.perf_counter() - t0) * 1e3 / LOOPS:.1f} ms per loop")
