"""In-situ duration of every kernel of the decode chain (WM_TIMELINE_FINE=1: a 1-thread stamp kernel behind every launch of
the chain writes the device clock, wm_debug_timeline), graph-replayed, the utterance groups sharing the chip as in the bench.
Each figure includes one stamp launch (~2-3 us); `alone` = the same with --groups 1 --batch <group size> (nothing beside it).
    python scripts/chain_probe.py [--batch 576] [--steps 6] [--groups 0] [--rows-path 1]"""
import argparse, os, sys, time
os.environ["WM_TIMELINE_FINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import numpy as np, torch
import bench, native, synthetic
from pathlib import Path
from decoding import WhisperDecoding
from encoding import WhisperEncoding

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=576); ap.add_argument("--steps", type=int, default=6); ap.add_argument("--groups", type=int, default=0)
ap.add_argument("--rows-path", type=int, default=1); ap.add_argument("--config", default="int8")
ap.add_argument("--beside-encoder", type=int, default=0, help="CUs of an encoder pass of the same batch running beside the loop (bench.py's pipelined steps)")
a = ap.parse_args()
args = argparse.Namespace(model="large-v2", config=a.config, seed=0, engine_cache="/tmp/wm_bench_engines")
eng = Path(args.engine_cache) / f"large-v2-{a.config}-seed0"
if not (eng / "decoder_config.json").exists():
    eng.parent.mkdir(parents=True, exist_ok=True); bench.build_engines(args, eng)
lib = native.load_library()
lib.wm_set_rows_path(40 if a.rows_path else 0)
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.micro_batches = a.groups or None
d = synthetic.DIMS["large-v2"]
mel = synthetic.synthetic_mel(a.batch, 2 * d["n_audio_ctx"], d["n_mels"], 1234).cuda()
xa = enc.get_audio_features_async(mel); torch.cuda.synchronize()
dec.detect_language(xa)
n_layer = dec.decoder_config["num_layers"]
cap = 8 * 16 * n_layer * (a.steps + 4)
buf = torch.zeros(1 + 3 * cap, dtype=torch.int64, device="cuda")
native.check(lib.wm_debug_timeline(buf.data_ptr(), cap))
dec.sample_len = a.steps
if a.beside_encoder:
    dec.main_loop(xa, ignore_eot=True); torch.cuda.synchronize()      # graph capture synchronises the device: capture first, measure afterwards
    buf.zero_()
    enc.prefetch(mel, a.beside_encoder)
    time.sleep(0.05)
t0 = time.perf_counter(); dec.main_loop(xa, ignore_eot=True)
if a.beside_encoder:
    enc.collect()
torch.cuda.synchronize()
native.check(lib.wm_debug_timeline(None, 0))
n = min(int(buf[0].item()) & 0xffffffff, cap)
ev = buf[1:1 + 3 * n].view(-1, 3).cpu().numpy()
names = {0: "layer start", 1: "qkv GEMM", 2: "self-attention", 3: "out GEMM", 4: "row kernel (out)", 5: "cq GEMM", "K0": "(to K/V launch)", "K1": "cross-attention K/V",
         6: "(post start)", 7: "cout GEMM", 8: "row kernel (cout)", 9: "mlp1 GEMM", 10: "row kernel (mlp1) | mlp1 GEMM fused", 11: "mlp2 GEMM", 12: "row kernel (mlp2)"}
acc = {}
for tag in sorted(set(ev[:, 0].tolist())):
    e = ev[ev[:, 0] == tag]
    e = e[np.argsort(e[:, 2], kind="stable")]
    # the last a.steps - 1 replayed steps only (prefill, capture and first replay dropped): cut at the layer-0 starts
    starts = [i for i in range(len(e)) if e[i, 1] == 1000]
    if len(starts) < 4: continue
    e = e[starts[-(a.steps - 2)]:]
    for (w0, t0_), (w1, t1_) in zip(e[:-1, 1:], e[1:, 1:]):
        k1 = ("K0" if w1 % 2 == 0 else "K1") if w1 < 1000 else int((w1 - 1000) % 32)
        if k1 == 0: continue                       # layer start: the gap to the previous layer's last stamp is the stamp itself
        acc.setdefault(k1, []).append((t1_ - t0_) / 100.0)
tot = 0.0
print(f"B={a.batch} groups={len(set(ev[:, 0].tolist()))} rows_path={a.rows_path} beside_encoder_cus={a.beside_encoder}: in-situ microseconds per launch (mean over layers, steps, groups; each includes one stamp launch)")
for k in [1, 2, 3, 4, 5, "K0", "K1", 6, 7, 8, 9, 10, 11, 12]:
    if k in acc:
        m = float(np.mean(acc[k])); tot += m
        print(f"  {names[k]:38s} {m:8.1f}   (p90 {float(np.percentile(acc[k], 90)):7.1f}, n={len(acc[k])})")
print(f"  sum per layer {tot:.1f} us -> {tot * n_layer / 1e3:.2f} ms per token step of one group")
