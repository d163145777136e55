"""Micro-benchmark: what the small-batch Linear (wm_gemv_fused, one row) gains when its weights are already in L2 / Infinity
Cache -- the ceiling of any scheme that prefetches the next Linear's weights.  A chain of launches captured in one hipGraph
(as the decode loop replays them), over (a) 64 different weight matrices of 5 MB each (cold: 320 MB, past every cache) and
(b) one matrix 64 times (warm)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
K = 1280
for N in (1280, 3840, 5120):
    n_mats = 64
    ws = [torch.randint(-127, 127, (N * K,), dtype=torch.int8, device="cuda") for _ in range(n_mats)]
    sc = torch.full((N,), 0.01, dtype=torch.float16, device="cuda")
    a = torch.randn(1, K, device="cuda").half()
    out = torch.zeros(1, N, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    def chain(mats):
        for w in mats:
            io = native.WmGemvIO()
            io.a, io.lda, io.m, io.k = a.data_ptr(), K, 1, K
            io.wt, io.n_blocks, io.w8 = w.data_ptr(), N // 16, 1
            io.scale, io.mode = sc.data_ptr(), 0
            io.out32, io.ld32 = out.data_ptr(), N
            native.check(lib.wm_gemv_fused(C.byref(io), torch.cuda.current_stream().cuda_stream))
    res = {}
    for name, mats in (("cold", ws), ("warm", [ws[0]] * n_mats)):
        with torch.cuda.stream(side):
            chain(mats); side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                chain(mats)
            g.replay(); side.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(10): g.replay()
            e1.record(side); side.synchronize()
            res[name] = e0.elapsed_time(e1) / (10 * n_mats) * 1e3
    print(f"N={N} K={K} int8, one row: {res['cold']:.2f} us per launch with cold weights, {res['warm']:.2f} us with the weights in cache", flush=True)
