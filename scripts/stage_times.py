"""Diagnostic: per-stage wall times of one batch on the GPU (engines from bench.py's cache)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
if os.environ.get("Q_BEFORE"): os.environ["GPU_MAX_HW_QUEUES"] = os.environ["Q_BEFORE"]
import torch
if os.environ.get("Q_AFTER"): os.environ["GPU_MAX_HW_QUEUES"] = os.environ["Q_AFTER"]
import bench, synthetic
from pathlib import Path
from decoding import WhisperDecoding
from encoding import WhisperEncoding

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--decode-steps", type=int, default=32)
ap.add_argument("--model", default="large-v2"); ap.add_argument("--config", default="int8"); ap.add_argument("--reps", type=int, default=2); ap.add_argument("--groups", type=int, default=0); ap.add_argument("--skip-enc", action="store_true"); ap.add_argument("--dummy-streams", type=int, default=0)
a = ap.parse_args()
args = argparse.Namespace(model=a.model, config=a.config, seed=0, engine_cache="/tmp/wm_bench_engines")
eng = Path(args.engine_cache) / f"{a.model}-{a.config}-seed0"
if not (eng / "decoder_config.json").exists():
    eng.parent.mkdir(parents=True, exist_ok=True)
    print("build s", bench.build_engines(args, eng))
dummies = [torch.cuda.Stream() for _ in range(a.dummy_streams)]
for ds in dummies:
    with torch.cuda.stream(ds):
        torch.zeros(16, device='cuda').add_(1)
torch.cuda.synchronize()
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.sample_len = a.decode_steps
dec.micro_batches = a.groups or None
d = synthetic.DIMS[a.model]
mel = synthetic.synthetic_mel(a.batch, 2 * d["n_audio_ctx"], d["n_mels"], 1234).cuda()
def T(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for rep in range(a.reps):
    xa, t_enc = T(lambda: enc.get_audio_features_async(mel))
    ckv, t_ckv = T(lambda: dec.xa2cross_key_value(xa)) if a.batch <= 256 else (None, 0.0)
    del ckv; torch.cuda.empty_cache()
    _, t_lang = T(lambda: dec.detect_language(xa))
    out, t_loop = T(lambda: dec.main_loop(xa, ignore_eot=True))
    print(f"rep {rep}: B={a.batch} enc {t_enc:.1f} ms ({t_enc/a.batch:.2f}/clip)  cross-KV {t_ckv:.1f} ms  lang-id {t_lang:.1f} ms  "
          f"decode loop {t_loop:.1f} ms = {t_loop/a.decode_steps:.3f} ms/step, {a.batch*a.decode_steps/t_loop*1e3:.0f} tok/s", flush=True)
print("tokens[0][:16]", out[0][0][:16].tolist(), "finite lp", bool(torch.isfinite(out[1]).all()))
