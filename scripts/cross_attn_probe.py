"""Runs the decode cross-attention kernel alone (kernel-level C-ABI entry) at the bench's launch shape, for
rocprofv3 --pmc FETCH_SIZE / --kernel-trace runs.  usage: cross_attn_probe.py [B=64] [reps=20] [peak=0] [skip=-1]
peak > 0: K rows scaled so that the score standard deviation is ~ peak (a sharply peaked softmax, as with real weights: most
probabilities round to fp16 zero); skip = 0 / 1: exact V-row skipping off / on (wm_set_cross_v_skip; -1 = the library's default)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, Tk = 20, 1500
peak = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
skip = int(sys.argv[4]) if len(sys.argv) > 4 else -1
lib.wm_set_cross_v_skip(skip)
kv = [torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(4)]       # 4 x 491 MB: defeats the 256 MB Infinity Cache
if peak > 0:
    for t in kv:
        t[:, 0] *= peak
q = torch.randn(B, H * 64, device="cuda")
out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)
s = torch.cuda.current_stream().cuda_stream
for r in range(reps):
    native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 4].data_ptr(), out.data_ptr(), 1, None, s))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(reps):
    native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 4].data_ptr(), out.data_ptr(), 1, None, s))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"B={B} peak={peak} skip={skip}: {us:.1f} us per launch, {B * H * 2 * Tk * 64 * 2 / us / 1e6:.2f} TB/s of algorithmic bytes; checksum {int(out.view(torch.int16).long().sum())}")
print("algorithmic bytes per launch", B * H * 2 * Tk * 64 * 2)
