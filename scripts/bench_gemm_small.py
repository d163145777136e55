"""Round 4: the encoder's GEMM shapes at FEW rows (1 .. 16 clips): 128 x 128 tiles on two workgroups per CU (gemm_f16.hip, NWAVE = 4)
against the persistent 256 x 256 kernel (gemm_f16p.hip), through the C ABI.  Prints both times and whether the outputs are
bit-identical (they must be: same MFMA chain per output element, same epilogue arithmetic).
    python scripts/bench_gemm_small.py [clips ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
clips = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 12, 16, 24]
shapes = [("qkv", 3840, 1280, 0, 0), ("out", 1280, 1280, 0, 1), ("mlp1", 5120, 1280, 1, 0), ("mlp2", 1280, 5120, 0, 1), ("ckv", 2560, 1280, 0, 0)]
REPS = int(os.environ.get("REPS", "20"))
s = torch.cuda.current_stream().cuda_stream
for B in clips:
    M = 1500 * B
    tot = {0: 0.0, 1: 0.0}
    for (name, N, K, act, res) in shapes:
        torch.manual_seed(N + K)
        A = (torch.randn(M, K, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        bias = torch.randn(N, device="cuda").half()
        R = torch.randn(M, N, device="cuda").half() if res else None
        outs, times = [], []
        for small in (0, 1):
            lib.wm_set_gemm_small_tiles(1 << 30 if small else 0)
            C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
            def run():
                native.check(lib.wm_gemm(A.data_ptr(), K, M, K, W.data_ptr(), N, 0, None, bias.data_ptr(),
                                         R.data_ptr() if res else None, N, act, C.data_ptr(), N, None, 0, s))
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS): run()
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / REPS * 1e3)
            outs.append(C)
        same = bool(torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)))
        tiles = ((M + 255) // 256) * (N // 256)
        if name != "ckv":
            tot[0] += times[0]; tot[1] += times[1]
        print(f"B={B:3d} {name:5s} M={M:6d} N={N} K={K} tiles256={tiles:5d}: persistent {times[0]:8.1f} us ({2*M*N*K/times[0]/1e6:5.0f} TF)  "
              f"small {times[1]:8.1f} us ({2*M*N*K/times[1]/1e6:5.0f} TF)  bit-identical {same}", flush=True)
    print(f"B={B:3d} layer GEMMs: persistent {tot[0]:.0f} us, small {tot[1]:.0f} us", flush=True)
lib.wm_set_gemm_small_tiles(-1)
