"""Micro-benchmark: one decode Linear at group sizes, the row-split fused kernel (wm_gemm_rows: LayerNorm prologue + epilogue
included) against the split-K weight-streaming GEMM alone (wm_gemm_skinny: its row kernel not included), chained launches on
one stream (hipGraph-free, back to back), large-v2 shapes, int8 weights.   python scripts/bench_rows.py [rows ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import numpy as np, torch, native
lib = native.load_library()
rows = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 192]
s = torch.cuda.current_stream().cuda_stream
K = 1280
for M in rows:
    for N in (1280, 3840, 5120):
        wt = torch.randint(-127, 128, (N // 16 * (K // 64) * 1024,), dtype=torch.int8, device="cuda")
        sc = (torch.rand(N, device="cuda") * 0.01).half()
        A = (torch.randn(M, K, device="cuda") * 0.5).half()
        g = torch.ones(K, device="cuda").half(); b = torch.zeros(K, device="cuda").half(); bias = torch.zeros(N, device="cuda").half()
        out32 = torch.empty(M, N, device="cuda"); out16 = torch.empty(M, N, device="cuda", dtype=torch.float16)
        x = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        ks = lib.wm_gemm_skinny_default_ksplit(M, K, N // 16, 1)
        part = torch.empty(ks, M, N, device="cuda")
        def rows_call(mode, ln):
            io = native.WmGemvIO()
            io.a, io.lda, io.m, io.k = A.data_ptr(), K, M, K
            io.wt, io.n_blocks, io.w8, io.scale = wt.data_ptr(), N // 16, 1, sc.data_ptr()
            io.mode, io.bias, io.gelu_kind = mode, bias.data_ptr(), 1
            io.out32, io.ld32, io.out16, io.ld16, io.n_valid, io.x, io.ldx = out32.data_ptr(), N, out16.data_ptr(), N, N, x.data_ptr(), N
            if ln: io.ln_gamma, io.ln_beta = g.data_ptr(), b.data_ptr()
            return io
        ios = {"rows m0+LN": rows_call(0, True), "rows m1+LN": rows_call(1, True), "rows m2": rows_call(2, False)}
        def timeit(f, n=200):
            for _ in range(20): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        res = {k: timeit(lambda io=io: native.check(lib.wm_gemm_rows(C.byref(io), s))) for k, io in ios.items()}
        res[f"skinny ks{ks}"] = timeit(lambda: native.check(lib.wm_gemm_skinny(A.data_ptr(), K, M, K, wt.data_ptr(), N // 16, 1, sc.data_ptr(), ks, part.data_ptr(), s)))
        print(f"M={M:4d} N={N:5d}: " + "  ".join(f"{k} {v:6.1f} us" for k, v in res.items()), flush=True)
