"""Micro-benchmark: aggregate bandwidth of 1 / 2 / 3 decode cross-attention launches running at the same time (one stream each,
group size B utterances each) -- what the K/V streams of the utterance groups get when they overlap in the decode loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
H, Tk = 20, 1500
B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
REPS = 48
kv = [[torch.randn(B, 2, H, Tk, 64, device="cuda").half() for _ in range(3)] for _ in range(3)]      # 3 streams x 3 buffers x 1.47 GB
q = torch.randn(B, H * 64, device="cuda")
outs = [torch.empty(B, H * 64, device="cuda", dtype=torch.float16) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
def run(g, r):
    native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[g][r % 3].data_ptr(), outs[g].data_ptr(), 1, None, streams[g].cuda_stream))
for n in (1, 2, 3):
    for r in range(6):
        for g in range(n): run(g, r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(REPS):
        for g in range(n): run(g, r)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = n * REPS * B * H * 2 * Tk * 64 * 2
    print(f"B={B}: {n} stream(s) at once: {dt / REPS * 1e6:.1f} us per round of {n} launch(es), aggregate {nbytes / dt / 1e12:.2f} TB/s", flush=True)
