"""Micro-benchmark: weight-streaming GEMM (decode shapes) through the C ABI; sweeps ksplit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import numpy as np, torch, native, weight as W
lib = native.load_library()
Ms = [int(x) for x in sys.argv[1:]] or [16, 32, 64]
s = torch.cuda.current_stream().cuda_stream
for M in Ms:
    for (N, K) in [(1280, 1280), (3840, 1280), (5120, 1280), (1280, 5120)]:
        q = torch.randint(-127, 127, (N, K), dtype=torch.int8)
        tiles = torch.from_numpy(W.tile_linear(q.numpy())).cuda()
        scale = torch.rand(N).half().cuda()
        A = torch.randn(M, K).half().cuda()
        kdef = lib.wm_gemm_skinny_default_ksplit(M, K, N // 16, 1)
        res = []
        for ks in sorted(set([1, 2, 4, 8, 10, 16, 20, kdef])):
            if ks > K // 64: continue
            part = torch.empty(ks, M, N, dtype=torch.float32, device="cuda")
            def run():
                native.check(lib.wm_gemm_skinny(A.data_ptr(), K, M, K, tiles.data_ptr(), N // 16, 1, scale.data_ptr(), ks, part.data_ptr(), s))
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            res.append(f"ks{ks}{'*' if ks == kdef else ''}:{e0.elapsed_time(e1) / 50 * 1e3:.1f}us")
        print(f"M={M} N={N} K={K} ({N*K/1e6:.1f} MB): " + " ".join(res), flush=True)
