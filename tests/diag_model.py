"""Diagnostic (not a test; lives under tests/ because it calls the oracle, which only test code may): per-config, per-step engine-vs-oracle differences on the micro model."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import synthetic
from test_gpu_model import build_engine
from decoding import WhisperDecoding
from encoding import WhisperEncoding
from oracle.whisper_oracle import *

dims = Dims(**synthetic.DIMS["micro"])
tmp = tempfile.mkdtemp()
for seed in (7, 8):
    sd = synthetic_state_dict(dims, seed)
    mel = synthetic_mel(2, 128, 80, 1234)
    for wo, i8 in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        scales = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=bool(wo))).calibrate_kv_scales(mel, 6) if i8 else None
        oracle = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=bool(wo), int8_kv=bool(i8), kv_scales=scales))
        ref = greedy_reference_run(oracle, mel, [5, 17, 900], 10)
        o32 = greedy_reference_run(OracleModel(dims, sd, OracleConfig(act="float32", weight_only=bool(wo), int8_kv=bool(i8), kv_scales=scales)), mel, [5, 17, 900], 10)
        eng = build_engine(tmp + f"/s{seed}", "micro", seed, bool(wo), bool(i8), scales)
        enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
        xa = enc.get_audio_features(mel.cuda())
        cross = dec.xa2cross_key_value(xa)
        logits, kv = dec.decode(torch.tensor([[5, 17, 900]] * 2).cuda(), cross)
        ds = [float((logits.float().cpu() - ref["logits"][0]).abs().max())]
        d32 = [float((ref["logits"][0] - o32["logits"][0]).abs().max())]
        for s in range(9):
            logits, kv = dec.decode(ref["ids"][:, s:s + 1].cuda(), cross, kv)
            ds.append(float((logits[:, 0].float().cpu() - ref["logits"][s + 1][:, 0]).abs().max()))
            d32.append(float((ref["logits"][s + 1] - o32["logits"][s + 1]).abs().max()))
        kvd = (kv[0].cpu().float() - ref["self_kv"][0].float()).abs()
        print(f"seed {seed} wo={wo} i8={i8} xa {float((xa.float().cpu()-ref['xa']).abs().max()):.4f} "
              f"logits(engine-oracle16) {['%.4f' % d for d in ds]} (oracle16-oracle32, free-running) max {max(d32):.4f} "
              f"kv0 maxdiff {float(kvd.max()):.4f} frac>0 {float((kvd>0).float().mean()):.4f} scales {scales}")
