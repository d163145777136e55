import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
os.environ["WM_GEMM_DBG_PTR"] = str(buf.data_ptr())
import native
lib = native.load_library()
M = 1500 * 128
names = {0: "late wait of tile 5: before", 1: "  after", 2: "epilogue entry", 3: "epilogue set-up done", 4: "all stores issued", 5: "next K loop starts",
         8: "stage 0 wait: before", 9: "  after", 10: "stage 1 wait: before", 11: "  after", 12: "stage 2 wait: before", 13: "  after", 14: "stage 3 wait: before", 15: "  after"}
for (N, K, act, res) in [(3840, 1280, 0, 0), (1280, 1280, 0, 1), (5120, 1280, 1, 0), (1280, 5120, 0, 1)]:
    A = (torch.randn(M, K, device="cuda") * 0.5).half(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    bias = torch.randn(N, device="cuda").half(); R = torch.randn(M, N, device="cuda").half() if res else None
    C = torch.empty(M, N, device="cuda", dtype=torch.float16); s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        native.check(lib.wm_gemm(A.data_ptr(), K, M, K, W.data_ptr(), N, 0, None, bias.data_ptr(), R.data_ptr() if res else None, N, act, C.data_ptr(), N, None, 0, s))
    torch.cuda.synchronize()
    b = buf.cpu().numpy()
    print(f"N={N} K={K} act={act} res={res}   (us since epilogue entry; wave 0 | wave 4)")
    for k in sorted(names):
        a0, a4 = (b[k] - b[2]) / 100.0, (b[32 + k] - b[32 + 2]) / 100.0
        print(f"   {names[k]:32s} {a0:9.2f} | {a4:9.2f}")
