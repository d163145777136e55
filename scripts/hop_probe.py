"""Diagnostic: cost of a cross-stream event dependency (hipEventRecord + hipStreamWaitEvent) between kernels,
plain and CU-masked streams."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
hip = C.CDLL("libamdhip64.so")
for f in ("hipEventCreateWithFlags", "hipEventRecord", "hipStreamWaitEvent", "hipStreamSynchronize"):
    getattr(hip, f).restype = C.c_int
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
torch.zeros(1, device="cuda")
counter = torch.zeros(1, dtype=torch.int32, device="cuda")
N = 2000
def events(n):
    out = []
    for _ in range(n):
        e = C.c_void_p(); assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0; out.append(e)
    return out
ev = events(2 * N)
def run(s1, s2, hops):
    hip.hipStreamSynchronize(s1); hip.hipStreamSynchronize(s2)
    t0 = time.perf_counter()
    for i in range(N):
        lib.wm_step_advance(counter.data_ptr(), s1)
        if hops:
            hip.hipEventRecord(ev[2 * i], s1); hip.hipStreamWaitEvent(s2, ev[2 * i], 0)
        lib.wm_step_advance(counter.data_ptr(), s2 if hops else s1)
        if hops:
            hip.hipEventRecord(ev[2 * i + 1], s2); hip.hipStreamWaitEvent(s1, ev[2 * i + 1], 0)
    th = time.perf_counter() - t0
    hip.hipStreamSynchronize(s1); hip.hipStreamSynchronize(s2)
    return th, time.perf_counter() - t0
plain = [torch.cuda.Stream() for _ in range(2)]
masked = [native.create_masked_stream([i < 64 for i in range(256)], 0), native.create_masked_stream([i >= 64 for i in range(256)], 1)]
for name, (a, b) in (("plain", plain), ("masked", masked)):
    for hops in (False, True):
        run(a.cuda_stream, b.cuda_stream, hops)
        th, t = run(a.cuda_stream, b.cuda_stream, hops)
        print(f"{name} hops={hops}: {t / (2 * N) * 1e6:.2f} us per kernel (host issue {th / (2 * N) * 1e6:.2f} us per kernel)", flush=True)
