"""Micro-benchmark: encoder self-attention kernel (wm_attn_encoder) at the large-v2 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
H, T = 20, 1500
for B in ((128, 256) if os.environ.get("ZERO_DATA") or os.environ.get("BIG_ONLY") else (1, 2, 4, 8, 32, 128)):
    qkv = (torch.randn(B * T, 3 * H * 64, device="cuda") * 0.5).half()
    if os.environ.get("ZERO_DATA"):       # all-zero operands: is the kernel held back by the clock the chip keeps under random-data MFMA power?
        qkv.zero_()
    out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.float16)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: native.check(lib.wm_attn_encoder(qkv.data_ptr(), 3 * H * 64, B, T, H, out.data_ptr(), H * 64, s))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"B={B}: {ms:.3f} ms ({ms / B * 1e3:.1f} us per clip-layer), {4.0 * T * T * 64 * H * B / ms / 1e9:.0f} TFLOP/s", flush=True)
