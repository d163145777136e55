"""Round 6: a variant library's encoder attention kernel against the product's (WM_LIBRARY_PATH=build/lab/libwm_<name>.so): a hash of the output on
fixed random inputs (bit-identity across processes) and the time at 128 / 256 clips, interleaved by the calling script."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
H, T = 20, 1500
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(7)
for B in (3, 128, 256):
    qkv = (torch.randn(B * T, 3 * H * 64, device="cuda", generator=g) * 0.5).half()
    out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.float16)
    run = lambda: native.check(lib.wm_attn_encoder(qkv.data_ptr(), 3 * H * 64, B, T, H, out.data_ptr(), H * 64, s))
    for _ in range(3): run()
    torch.cuda.synchronize()
    if B == 3:
        print("hash", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{os.environ.get('WM_LIBRARY_PATH', 'product')}: B={B}: {ms:.3f} ms, {4.0 * T * T * 64 * H * B / ms / 1e9:.0f} TFLOP/s", flush=True)
