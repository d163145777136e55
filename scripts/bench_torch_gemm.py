"""Reference point: what the vendor GEMM library (hipBLASLt through torch.matmul) reaches on the encoder's GEMM shapes.
Not used by the engine (its GEMMs fuse bias / GELU / residual / head split); this only says how much headroom gemm_f16 has."""
import torch, time
M = 192000
for (K, N) in ((1280, 3840), (1280, 1280), (1280, 5120), (5120, 1280)):
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16)
    for _ in range(3): c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"torch.matmul M={M} K={K} N={N}: {ms:.3f} ms, {2 * M * K * N / ms / 1e9:.0f} TFLOP/s", flush=True)
