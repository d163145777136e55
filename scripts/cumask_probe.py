"""Diagnostic: how much does a latency-bound kernel chain (weight-streaming GEMM + row kernel, what a decode
step is made of between its cross-attention kernels) slow down while ANOTHER stream streams cross K/V at
HBM speed, and does pinning the two streams to disjoint CU sets (hipExtStreamCreateWithCUMask) help?
usage: cumask_probe.py [light_cus=64]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import numpy as np, torch, native, weight as W
lib = native.load_library()
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]

def plain_stream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; return s

def masked_stream(cu_pred):
    """cu_pred(i) -> bool for CU index i in [0, 256)"""
    words = (C.c_uint32 * 8)()
    for i in range(256):
        if cu_pred(i): words[i // 32] |= (1 << (i % 32))
    s = C.c_void_p(); rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words); assert rc == 0, rc; return s

torch.cuda.init(); torch.zeros(1, device="cuda")
B, H, Tk, M, K, N = 128, 20, 1500, 128, 1280, 1280
kv = [torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(2)]
q = torch.randn(B, H * 64, device="cuda")
out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)
qw = torch.randint(-127, 127, (N, K), dtype=torch.int8)
tiles = torch.from_numpy(W.tile_linear(qw.numpy())).cuda()
scale = torch.rand(N).half().cuda()
A = torch.randn(M, K).half().cuda()
ks = lib.wm_gemm_skinny_default_ksplit(M, K, N // 16, 1)
part = torch.empty(ks, M, N, dtype=torch.float32, device="cuda")
g = torch.ones(N, device="cuda").half(); bta = torch.zeros(N, device="cuda").half()
x = torch.randn(M, N, device="cuda").half(); xn = torch.empty_like(x)
torch.cuda.synchronize()

def heavy(s, n):
    for r in range(n):
        native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r % 2].data_ptr(), out.data_ptr(), 1, None, s))

def light(s, n):
    for _ in range(n):
        native.check(lib.wm_gemm_skinny(A.data_ptr(), K, M, K, tiles.data_ptr(), N // 16, 1, scale.data_ptr(), ks, part.data_ptr(), s))
        native.check(lib.wm_layernorm(x.data_ptr(), N, M, N, g.data_ptr(), bta.data_ptr(), xn.data_ptr(), N, s))

def timed(fn_list):
    for s, _, _ in fn_list: hip.hipStreamSynchronize(s)
    t0 = time.perf_counter()
    for s, fn, n in fn_list: fn(s, n)
    ends = []
    for s, _, _ in fn_list:
        hip.hipStreamSynchronize(s); ends.append(time.perf_counter() - t0)
    return ends

light_cus = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NL, NH = 2000, 200
hip.hipStreamCreateWithPriority.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_int]
def prio_stream(p):
    s = C.c_void_p(); assert hip.hipStreamCreateWithPriority(C.byref(s), 1, p) == 0; return s
lo, hi = C.c_int(), C.c_int(); hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)); print("priority range (least, greatest):", lo.value, hi.value)
configs = {"plain": (plain_stream(), plain_stream()), "light high priority": (prio_stream(lo.value), prio_stream(hi.value))}
# CU index -> (XCD = i % 8?) unknown mapping: try both "first k CUs" and "k/8 CUs of every XCD" readings
configs[f"mask low {light_cus} CUs light"] = (masked_stream(lambda i: i >= light_cus), masked_stream(lambda i: i < light_cus))
configs[f"mask strided {light_cus} CUs light"] = (masked_stream(lambda i: (i % 32) >= light_cus // 8), masked_stream(lambda i: (i % 32) < light_cus // 8))
for name, (sh, sl) in configs.items():
    heavy(sh, 4); light(sl, 20)
    th = timed([(sh, heavy, NH)])[0]
    tl = timed([(sl, light, NL)])[0]
    both = timed([(sh, heavy, NH), (sl, light, NL)])
    print(f"{name}: heavy alone {th / NH * 1e6:.1f} us/launch ({B * H * 2 * Tk * 64 * 2 / (th / NH) / 1e12:.2f} TB/s), "
          f"light alone {tl / NL * 1e6:.2f} us/pair; together: heavy done {both[0] * 1e3:.1f} ms, light done {both[1] * 1e3:.1f} ms "
          f"(serial sum {(th + tl) * 1e3:.1f} ms)", flush=True)
