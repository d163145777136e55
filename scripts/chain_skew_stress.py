"""Round 6: is the give-up of the one-launch step beside the START of an encoder pass a residency problem or a race inside the step?
Take the encoder away and keep the skew: short occupancies (wm_debug_occupy: N workgroups holding LDS for a few hundred microseconds, so the step's
workgroups on those CUs start late) issued on a side stream right before / during ragged decode loops of 5 utterances.  A step whose workgroups
merely start late must finish (later); a give-up here is a race in the kernel.
    python scripts/chain_skew_stress.py [loops] [occupied CUs] [microseconds] [live 0|1]"""
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
LOOPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N_WG = int(sys.argv[2]) if len(sys.argv) > 2 else 96
US = int(sys.argv[3]) if len(sys.argv) > 3 else 300
LIVE = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
B = int(os.environ.get("B", "5"))
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
lib = native.load_library()
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.sample_len = 128
dec.skip_finished_rows = LIVE
g = torch.Generator(device="cuda").manual_seed(1234)
mel = (torch.randn((B, 80, 3000), generator=g, device="cuda") * 0.5).clamp_(-0.5, 1.5).half()
limits_all = bench.librispeech_like_lengths(5 * B, 128)
deal = np.asarray(dec.balanced_order(B))
batches = [limits_all[k * B:(k + 1) * B][deal] for k in range(5)]
xa = enc.get_audio_features_async(mel)
dec.detect_language(xa); dec.main_loop(xa, row_limit=torch.as_tensor(batches[0], dtype=torch.int32)); torch.cuda.synchronize()
side = torch.cuda.Stream()
gave_up = 0
t0 = time.perf_counter()
for it in range(LOOPS):
    limits = batches[it % 5]
    dec.detect_language(xa)
    for k in range(24):            # ~ 24 x US microseconds of occupancies queued on the side stream: they run beside the loop's first steps
        native.check(lib.wm_debug_occupy(N_WG, 100 * 1024, US, side.cuda_stream), "wm_debug_occupy")
    dec.main_loop(xa, row_limit=torch.as_tensor(limits, dtype=torch.int32))
    torch.cuda.synchronize()
    st = native.chain_status()
    if st["declined"] or st["error_pending"]:
        gave_up += 1
        print(f"  loop {it} limits {list(limits)}: GAVE UP ({st['reason']})", flush=True)
        err = C.c_int(0); lib.wm_decode_chain_error(C.byref(err)); lib.wm_set_decode_chain(-1)
print(f"B={B} live={LIVE} occupancies of {N_WG} workgroups x {US} us: {gave_up} give-ups in {LOOPS} loops, {(time.perf_counter() - t0) * 1e3 / LOOPS:.1f} ms per loop; chain launches {native.chain_status()['launches']}")
