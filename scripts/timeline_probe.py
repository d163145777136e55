"""Diagnostic: device-side timeline of the decode loop's cross-attention launches per utterance group (wm_debug_timeline).
Prints, per group, the mean duration of the K/V launches, the mean gap between them (the group's short-kernel chain), and how
many K/V launches of different groups overlap in time.  usage: timeline_probe.py [batch=576] [groups=0 (auto)] [steps=12]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import argparse, numpy as np, torch
import bench, synthetic, native
from pathlib import Path
from decoding import WhisperDecoding
from encoding import WhisperEncoding
B = int(sys.argv[1]) if len(sys.argv) > 1 else 576
G = int(sys.argv[2]) if len(sys.argv) > 2 else 0
T = int(sys.argv[3]) if len(sys.argv) > 3 else 12
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
if not (eng / "decoder_config.json").exists():
    eng.parent.mkdir(parents=True, exist_ok=True)
    bench.build_engines(argparse.Namespace(model="large-v2", config="int8", seed=0), eng)
lib = native.load_library()
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.micro_batches = G or None
mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
xa = enc.get_audio_features_async(mel)
dec.detect_language(xa)
cap = 3 * 32 * 2 * (T + 4) * 8
buf = torch.zeros(1 + 3 * cap, dtype=torch.int64, device="cuda")
native.check(lib.wm_debug_timeline(buf.data_ptr(), cap))       # BEFORE the graphs are captured: the stamps are part of them
dec.sample_len = T
dec.main_loop(xa, ignore_eot=True)                              # captures
buf.zero_()
dec.main_loop(xa, ignore_eot=True)
torch.cuda.synchronize()
native.check(lib.wm_debug_timeline(None, 0))
n = int(buf[0].item() & 0xffffffff)
ev = buf[1:1 + 3 * min(n, cap)].view(-1, 3).cpu().numpy()
tags = sorted(set(ev[:, 0].tolist()))
print(f"B={B} groups={len(tags)} entries={n}")
spans = []
for gi, tag in enumerate(tags):
    e = ev[ev[:, 0] == tag]
    e = e[np.argsort(e[:, 2], kind="stable")]
    starts = e[e[:, 1] % 2 == 0][:, 2]; ends = e[e[:, 1] % 2 == 1][:, 2]
    m = min(len(starts), len(ends)); starts, ends = starts[:m], ends[:m]
    dur = (ends - starts) / 100.0                                 # us
    gap = (starts[1:] - ends[:-1]) / 100.0
    keep = slice(64, None)                                        # skip the eager prefill + first steps
    print(f"  group {gi}: {m} launches, K/V launch {np.mean(dur[keep]):7.1f} us (p10 {np.percentile(dur[keep], 10):6.1f}, p90 {np.percentile(dur[keep], 90):6.1f}), "
          f"chain between launches {np.mean(gap[keep]):7.1f} us (p10 {np.percentile(gap[keep], 10):6.1f}, p90 {np.percentile(gap[keep], 90):6.1f})")
    spans += [(s, 1) for s in starts[keep]] + [(t, -1) for t in ends[keep]]
spans.sort()
level, last, acc = 0, None, {}
for t, d in spans:
    if last is not None: acc[level] = acc.get(level, 0) + (t - last)
    level += d; last = t
tot = sum(acc.values())
print("  time with k K/V launches in flight: " + ", ".join(f"{k}: {100.0 * v / tot:.1f} %" for k, v in sorted(acc.items())))
