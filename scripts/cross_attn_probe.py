"""Runs the decode cross-attention kernel alone (kernel-level C-ABI entry) at the bench's launch shape, for
rocprofv3 --pmc FETCH_SIZE / --kernel-trace runs.  usage: cross_attn_probe.py [B=64] [reps=20]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, Tk = 20, 1500
kv = [torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(4)]       # 4 x 491 MB: defeats the 256 MB Infinity Cache
q = torch.randn(B, H * 64, device="cuda")
out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)
s = torch.cuda.current_stream().cuda_stream
for r in range(reps):
    native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 4].data_ptr(), out.data_ptr(), 1, None, s))
torch.cuda.synchronize()
print("algorithmic bytes per launch", B * H * 2 * Tk * 64 * 2)
