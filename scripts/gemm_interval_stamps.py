"""Diagnostic: where a K stage of the persistent GEMM (gemm_f16p.hip) spends its time, interval by interval.  Needs a library
built with -DWM_GEMM_STAMPS=<point mask> (scripts/call55.sh builds it next to the product library and passes its path):
core-clock stamps of waves 0 and 4 of one workgroup at the boundaries of 16 consecutive K stages of its second tile.
Points: 0 after barrier 1 (first multiply starts) | 1 its 16 MFMAs issued | 2 after barrier 2 | 3 second-half fragment reads and DMA
requests issued | 4 stage wait (vmcnt) done | 5 after barrier 3 (second multiply starts) | 6 its MFMAs issued | 7 after barrier 4 | 8 / 9 a second stamp right behind 0 / 5 (the first stamp's wait absorbs the wait for the
fragment reads: 8 -> 1 and 9 -> 6 are the MFMA issue alone)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import numpy as np, torch, native
lib = native.load_library()
mask = int(os.environ.get("STAMP_MASK", "0"), 0)
stamped = hasattr(lib, "wm_debug_gemm_stamps") and mask != 0
M = 1500 * int(os.environ.get("CLIPS", "128"))
for (N, K, act, res) in [(3840, 1280, 0, 0), (1280, 5120, 0, 1)]:
    torch.manual_seed(N + K)
    A = (torch.randn(M, K, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    bias = torch.randn(N, device="cuda").half()
    R = torch.randn(M, N, device="cuda").half() if res else None
    Cm = torch.empty(M, N, device="cuda", dtype=torch.float16)
    s = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros(320, dtype=torch.int64, device="cuda")
    if stamped:
        lib.wm_debug_gemm_stamps.argtypes = [C.c_void_p]
        assert lib.wm_debug_gemm_stamps(buf.data_ptr()) == 0
    def run():
        native.check(lib.wm_gemm(A.data_ptr(), K, M, K, W.data_ptr(), N, 0, None, bias.data_ptr(),
                                 R.data_ptr() if res else None, N, act, Cm.data_ptr(), N, None, 0, s))
    for _ in range(20): run()                    # the chip settles at the clock it holds under this load
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"M={M} N={N} K={K} act={act} res={res} mask={mask:#x}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
    if not stamped:
        continue
    st = buf.cpu().numpy().reshape(2, 16, 10).astype(np.int64)
    order = [0, 8, 1, 2, 3, 4, 5, 9, 6, 7]       # program order of the points inside a stage
    pts = [p for p in order if (mask >> p) & 1]
    for w in range(2):
        t = st[w][:, pts]                       # [stage, point]
        if (t == 0).any():
            print(f"  wave {4 * w}: stamps missing"); continue
        seg = np.diff(np.concatenate([t, np.roll(t[:, :1], -1, axis=0)], axis=1), axis=1)[:-1]       # point -> next point (last: -> next stage's first)
        stage = np.diff(t[:, 0])
        names = [f"{a}->{b}" for a, b in zip(pts, pts[1:] + [f"{pts[0]}'"])]
        print(f"  wave {4 * w}: stage {stage.mean():7.0f} cycles (min {stage.min()}, max {stage.max()});  " +
              "  ".join(f"{n} {seg[:, i].mean():6.0f}" for i, n in enumerate(names)))
