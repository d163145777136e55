"""Micro-benchmark: the decode cross-attention kernel alone at the decode loop's group size (B utterances x 20 heads x 1500 keys)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
H, Tk = 20, 1500
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = torch.cuda.current_stream().cuda_stream
kv = [torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(6)]       # 6 x 49 MB x B/128: well past the caches
q = torch.randn(B, H * 64, device="cuda")
out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)
def run(r):
    native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 6].data_ptr(), out.data_ptr(), 1, None, s))
for r in range(12): run(r)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(96): run(r)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 96 * 1e3
print(f"B={B} persist={os.environ.get('WM_CROSS_PERSIST_WGS', 'default')} rot={os.environ.get('WM_CROSS_ROT', '0')}: {us:.1f} us/launch, "
      f"{B * H * 2 * Tk * 64 * 2 / us / 1e6:.2f} TB/s", flush=True)
