"""Diagnostic: cross-attention kernel bandwidth vs number of CUs its stream may use (low-bit CU masks)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
B, H, Tk = 128, 20, 1500
kv = [torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(2)]
q = torch.randn(B, H * 64, device="cuda")
out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)
torch.cuda.synchronize()
for n in (32, 64, 96, 128, 160, 192, 224, 256):
    for kind in ("low", "high"):
        mask = [(i < n) if kind == "low" else (i >= 256 - n) for i in range(256)]
        s = native.create_masked_stream(mask, 0)
        for r in range(4):
            native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 2].data_ptr(), out.data_ptr(), 1, None, s.cuda_stream))
        s.synchronize(); t0 = time.perf_counter()
        for r in range(40):
            native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 2].data_ptr(), out.data_ptr(), 1, None, s.cuda_stream))
        s.synchronize(); t = (time.perf_counter() - t0) / 40
        print(f"{kind} {n:3d} CUs: {t * 1e6:.1f} us/launch, {B * H * 2 * Tk * 64 * 2 / t / 1e12:.2f} TB/s ({B * H * 2 * Tk * 64 * 2 / t / 1e9 / n:.1f} GB/s per CU)", flush=True)
