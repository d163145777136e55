"""Build-time check of gemm_f16p.hip's generated code (no GPU needed).  The first stage wait of a tile leaves the previous tile's
epilogue stores in flight by COUNT (N_STORES in the kernel): 16 in the SIMPLE kernels (16-byte stores), 32 in the general ones
(8-byte / 4-byte stores).  The count must not exceed what an epilogue copy really issues, so: every SIMPLE kernel holds 16-byte stores
only, 16 per epilogue copy (3 copies without an activation: residual / column scale / plain; 1 with GELU), and the general kernels
hold no 16-byte store at all.      python scripts/check_gemm_isa.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "eddie-wang-hackathon2023_amd", "csrc", "gemm_f16p.hip")
out = os.path.join(tempfile.mkdtemp(), "gemm_f16p.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
bad = 0
for m in re.finditer(r"^(_ZN2wm16gemm_f16p_kernelILi4ELi(\d)ELb(\d)EEEvNS_13GemmBigParamsE):.*?s_endpgm", text, re.S | re.M):
    name, act, simple, body = m.group(1), int(m.group(2)), m.group(3) == "1", m.group(0)
    x4, x2 = len(re.findall(r"global_store_dwordx4\s", body)), len(re.findall(r"global_store_dwordx2\s", body))
    ok = (x2 == 0 and x4 == (48 if act == 0 else 16)) if simple else (x4 == 0 and x2 > 0 and x2 % 32 == 0)
    print(("ok   " if ok else "BAD  ") + name, f"16-byte stores {x4}, 8-byte stores {x2}")
    bad += not ok
sys.exit(1 if bad else 0)
