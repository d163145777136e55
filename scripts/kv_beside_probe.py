"""What slows the decode cross-attention stream beside the encoder?  The K/V kernel (192 utterances per launch, the bench's
group) timed alone and while ONE encoder kernel type runs on a CU budget on another stream:
    python scripts/kv_beside_probe.py            (WM_GEMM_MAX_WGS / WM_ATTN_MAX_WGS set per case by this script's children)
Each case is a child process (the lab knobs are read once per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
    import torch, native
    lib = native.load_library()
    case = sys.argv[2]
    H, Tk, B = 20, 1500, 192
    kv = [torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(4)]
    q = torch.randn(B, H * 64, device="cuda")
    out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)
    side = torch.cuda.Stream()
    main = torch.cuda.Stream()
    M = 1500 * 256
    def gemm_inputs(N, K):
        A = (torch.randn(M, K, device="cuda") * 0.5).half(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        return A, W, torch.randn(N, device="cuda").half(), torch.empty(M, N, device="cuda", dtype=torch.float16)
    bg = None
    if case.startswith("gemm"):
        A, W, bias, C = gemm_inputs(3840, 1280)
        bg = lambda: native.check(lib.wm_gemm(A.data_ptr(), 1280, M, 1280, W.data_ptr(), 3840, 0, None, bias.data_ptr(), None, 0, 0, C.data_ptr(), 3840, None, 0, side.cuda_stream))
    elif case.startswith("attn"):
        Bq = 64
        qkv = (torch.randn(Bq * 1500, 3 * H * 64, device="cuda") * 0.5).half(); ao = torch.empty(Bq * 1500, H * 64, device="cuda", dtype=torch.float16)
        bg = lambda: native.check(lib.wm_attn_encoder(qkv.data_ptr(), 3 * H * 64, Bq, 1500, H, ao.data_ptr(), H * 64, side.cuda_stream))
    elif case.startswith("ln"):
        x = torch.randn(M, 1280, device="cuda").half(); g = torch.ones(1280, device="cuda").half(); y = torch.empty_like(x)
        bg = lambda: native.check(lib.wm_layernorm(x.data_ptr(), 1280, M, 1280, g.data_ptr(), g.data_ptr(), y.data_ptr(), 1280, side.cuda_stream))
    def kv_run(n):
        for r in range(n):
            native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 4].data_ptr(), out.data_ptr(), 1, None, main.cuda_stream))
    kv_run(4); torch.cuda.synchronize()
    t_bg = None
    if bg is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        bg(); torch.cuda.synchronize()
        e0.record(side); bg(); e1.record(side); torch.cuda.synchronize(); t_bg = e0.elapsed_time(e1)
        for _ in range(6): bg()                       # keep the side stream busy for the whole measurement
    import time; time.sleep(0.002)
    n = 12
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main); kv_run(n); e1.record(main)
    main.synchronize()
    busy = not side.query() if bg is not None else None
    ms = e0.elapsed_time(e1) / n
    torch.cuda.synchronize()
    print(f"{case:28s} K/V launch {ms * 1e3:7.1f} us = {B * H * 2 * Tk * 64 * 2 / ms / 1e9:6.2f} TB/s"
          + (f"   (background kernel alone: {t_bg:.2f} ms per launch; still running at the end: {busy})" if bg is not None else ""), flush=True)
    sys.exit(0)
cases = [("alone", {}), ("gemm_full_chip", {}), ("gemm_96", {"WM_GEMM_MAX_WGS": "96"}), ("gemm_64", {"WM_GEMM_MAX_WGS": "64"}), ("gemm_128", {"WM_GEMM_MAX_WGS": "128"}),
         ("attn_full_chip", {}), ("attn_192", {"WM_ATTN_MAX_WGS": "192"}), ("ln_stream", {})]
for name, env in cases:
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", name], env=dict(os.environ, **env))
