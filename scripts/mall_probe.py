"""Round 6 probe: what the decode cross-attention kernel gains when its K/V rows are already in the Infinity Cache (256 MiB).
For B rows (a layer's K/V = B x 7.68 MB): the launch timed (HIP events around the launch alone)
  cold      32 distinct K/V buffers in turn (the decode loop's situation: 32 layers, B x 245.76 MB per step)
  warm      the same buffer again and again
  touched   cold buffers, each read once (torch .sum over the bytes) right before its launch
    python scripts/mall_probe.py > gpurun_out/r6b_mall_probe.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, native
lib = native.load_library()
H, Tk = 20, 1500
s = torch.cuda.current_stream().cuda_stream
for B in (4, 8, 16, 24, 32):
    NL = 32
    kv = [torch.randn(B, 2, H, Tk, 64, device="cuda", dtype=torch.float16) for _ in range(NL)]
    q = torch.randn(B, H * 64, device="cuda")
    out = torch.empty(B, H * 64, device="cuda", dtype=torch.float16)
    ns = 4 if B * H <= 160 else 1
    ws = torch.empty(B * H * ns * 66, device="cuda")
    def launch(r):
        native.check(lib.wm_attn_decode_cross(q.data_ptr(), B, 1, H, Tk, kv[r].data_ptr(), out.data_ptr(), ns, ws.data_ptr(), s))
    def timed(mode, reps=3):
        tot, n = 0.0, 0
        for rep in range(reps):
            for r in range(NL):
                idx = 0 if mode == "warm" else r
                if mode == "touched":
                    kv[idx].view(torch.int32).sum()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); launch(idx); e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    tot += e0.elapsed_time(e1); n += 1
        return tot / n * 1e3
    mb = B * 7.68
    res = {m: timed(m) for m in ("cold", "warm", "touched")}
    print(f"B={B:3d} ({mb:6.1f} MB per launch, nsplit {ns}): " + "  ".join(f"{m} {t:6.1f} us = {mb / t:5.2f} TB/s" for m, t in res.items()), flush=True)
    del kv
    torch.cuda.empty_cache()
