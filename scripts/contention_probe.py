"""Diagnostic: which of the decode step's short kernels slows down, and by how much, while another hardware queue
streams cross-attention K/V?  Each kernel type is launched as a chain of N back-to-back launches on its own queue,
alone and next to a continuously running cross-attention stream (both queues dedicated, as the decode loop's are).
The 1-thread `wm_step_advance` is the null kernel: its chain time is pure launch + dispatch latency."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native, weight as W
lib = native.load_library()
torch.zeros(1, device="cuda")
sh = native.create_masked_stream([True] * 256, 0)
sl = native.create_masked_stream([True] * 256, 1)
B, H, Tk, M, C_ = 128, 20, 1500, 128, 1280
kv = [torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(3)]
q = torch.randn(B, H * 64, device="cuda")
out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)

def mk_gemm(K, N):
    qw = torch.randint(-127, 127, (N, K), dtype=torch.int8)
    tiles = torch.from_numpy(W.tile_linear(qw.numpy())).cuda()
    scale = torch.rand(N).half().cuda()
    A = torch.randn(M, K).half().cuda()
    ks = lib.wm_gemm_skinny_default_ksplit(M, K, N // 16, 1)
    part = torch.empty(ks, M, N, dtype=torch.float32, device="cuda")
    def run(s):
        native.check(lib.wm_gemm_skinny(A.data_ptr(), K, M, K, tiles.data_ptr(), N // 16, 1, scale.data_ptr(), ks, part.data_ptr(), s))
    run.keep = (tiles, scale, A, part)
    return run

g = torch.ones(C_, device="cuda").half(); bta = torch.zeros(C_, device="cuda").half()
x = torch.randn(M, C_, device="cuda").half(); xn = torch.empty_like(x)
def run_ln(s):
    native.check(lib.wm_layernorm(x.data_ptr(), C_, M, C_, g.data_ptr(), bta.data_ptr(), xn.data_ptr(), C_, s))
counter = torch.zeros(1, dtype=torch.int32, device="cuda")
def run_null(s):
    native.check(lib.wm_step_advance(counter.data_ptr(), s))
cap, T = 448, 64
qkv = torch.randn(M, 3 * C_, device="cuda")
cache = torch.zeros(M, 2, H, cap, 64, dtype=torch.int8, device="cuda")
ctx = torch.empty(M, C_, device="cuda", dtype=torch.float16)
def run_self(s):
    native.check(lib.wm_attn_decode_self(qkv.data_ptr(), M, 1, T, H, cache.data_ptr(), cap, cache.data_ptr(), cap, 1, C.c_float(0.05), ctx.data_ptr(), s))

kernels = {"null (1 thread)": run_null, "layernorm 128x1280": run_ln, "gemm 1280->1280": mk_gemm(1280, 1280),
           "gemm 1280->3840": mk_gemm(1280, 3840), "gemm 1280->5120": mk_gemm(1280, 5120), "gemm 5120->1280": mk_gemm(5120, 1280),
           "self-attention T=64": run_self}

def heavy(n):
    for r in range(n):
        native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 3].data_ptr(), out.data_ptr(), 1, None, sh.cuda_stream))

N = 1500
heavy(8); torch.cuda.synchronize()
t0 = time.perf_counter(); heavy(200); sh.synchronize(); th = (time.perf_counter() - t0) / 200
print(f"cross-attention alone: {th * 1e6:.1f} us/launch ({B * H * 2 * Tk * 64 * 2 / th / 1e12:.2f} TB/s)", flush=True)
for name, fn in kernels.items():
    for _ in range(20): fn(sl.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn(sl.cuda_stream)
    sl.synchronize(); alone = (time.perf_counter() - t0) / N
    n_heavy = int(alone * N * 4 / th) + 200           # enough K/V launches to outlast a 4x slower chain
    heavy(n_heavy)
    time.sleep(0.002)
    t0 = time.perf_counter()
    for _ in range(N): fn(sl.cuda_stream)
    sl.synchronize(); both = (time.perf_counter() - t0) / N
    busy = not sh.query()
    sh.synchronize()
    print(f"{name:<22} alone {alone * 1e6:6.2f} us   next to the K/V stream {both * 1e6:6.2f} us  ({both / alone:.2f}x)"
          + ("" if busy else "   [K/V stream ended first: lower bound]"), flush=True)
