"""Round 6: does the one-launch step survive bench.py's second figure at small batches?  For B = 3..8: five ragged batches (LibriSpeech-like
row limits, live-row lists) one stage after the other, then pipelined (the next batch's encoder on 96 CUs beside the loop), repeated; after
every phase the chain status (a give-up = 1 s stall + re-decode + the device off the form).    python scripts/chain_ragged_stress.py [repeats]"""
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
REPEATS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
BATCHES = [int(x) for x in os.environ.get("BATCHES", "3 4 5 6 7 8").split()]
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
lib = native.load_library()
for B in BATCHES:
    enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
    dec.sample_len = 128
    g = torch.Generator(device="cuda").manual_seed(1234)
    mel = (torch.randn((B, 80, 3000), generator=g, device="cuda") * 0.5).clamp_(-0.5, 1.5).half()
    limits_all = bench.librispeech_like_lengths(5 * B, 128)
    deal = np.asarray(dec.balanced_order(B))
    batches = [limits_all[k * B:(k + 1) * B][deal] for k in range(5)]
    for rep in range(REPEATS):
        for mode in ("sequential", "pipelined"):
            lib.wm_set_decode_chain(-1)
            err = C.c_int(0); lib.wm_decode_chain_error(C.byref(err)); lib.wm_set_decode_chain(-1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            xa = enc.get_audio_features_async(mel)
            for i, limits in enumerate(batches):
                dec.detect_language(xa)
                if mode == "pipelined" and i + 1 < len(batches):
                    enc.prefetch(mel, 96)
                dec.main_loop(xa, row_limit=torch.as_tensor(limits, dtype=torch.int32))
                if mode == "pipelined" and i + 1 < len(batches):
                    enc.loop_ended(); xa = enc.collect()
                else:
                    xa = enc.get_audio_features_async(mel)
                st = native.chain_status()
                if st["declined"] or st["error_pending"]:
                    print(f"  B={B} rep {rep} {mode} batch {i} limits {list(limits)}: GAVE UP: {st['reason']}", flush=True)
                    lib.wm_decode_chain_error(C.byref(err)); lib.wm_set_decode_chain(-1)
            torch.cuda.synchronize()
            print(f"B={B} rep {rep} {mode}: {(time.perf_counter() - t0) * 1e3 / 5:.1f} ms per batch, chain launches {native.chain_status()['launches']}", flush=True)
    del enc, dec
    torch.cuda.empty_cache()
