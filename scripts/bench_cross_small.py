"""Micro-benchmark: decode cross-attention at small batches, key-split factor sweep (kernel + combine)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
H, Tk = 20, 1500
s = torch.cuda.current_stream().cuda_stream
for B in (1, 2, 4, 8, 16, 24):
    kv = [torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(8)]
    q = torch.randn(B, H * 64, device="cuda")
    out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)
    res = []
    for ns in (1, 2, 4, 8, 16):
        ws = torch.empty(B * H * ns * 66, device="cuda")
        def run(r):
            native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 8].data_ptr(), out.data_ptr(), ns, ws.data_ptr(), s))
        for r in range(8): run(r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(64): run(r)
        e1.record(); torch.cuda.synchronize()
        res.append(f"ns{ns}:{e0.elapsed_time(e1) / 64 * 1e3:.1f}us")
    print(f"B={B}: " + " ".join(res), flush=True)
