"""Diagnostic: where the language-ID stage of one batch goes (cross-K/V projection, one-token decoder pass, host-side post-processing)."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 576
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
def T(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for rep in range(3):
    xa, t_enc = T(lambda: enc.get_audio_features_async(mel))
    st = dec._fast_state(B, xa.device)
    _, t_ckv = T(lambda: dec._cross_persistent(xa, st))
    _, t_lang = T(lambda: dec.detect_language(xa))                      # cross K/V cached: the pass itself + host work
    lg = st['lang_logits'][:, 0].float()
    _, t_host = T(lambda: dec._language_from_logits(lg, B, False))
    print(f"rep {rep}: B={B} encoder {t_enc:.1f} ms | cross-K/V projection {t_ckv:.1f} ms | language pass (decoder + host) {t_lang:.1f} ms, of which host-side _language_from_logits {t_host:.1f} ms", flush=True)
