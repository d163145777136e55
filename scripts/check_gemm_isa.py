"""Build-time check of gemm_f16p.hip's generated code (no GPU needed): the SIMPLE kernels (the encoder layers' four GEMMs) store 16 bytes
per lane and instruction -- 16 stores per epilogue copy (4 copies without an activation: head-split / residual / column scale / plain; 1 with GELU) --
and the general kernels 8 bytes (a multiple of 32 per copy); no SIMPLE kernel and no kernel of the encoder's default configuration
(ACT 0 / 1) spills a register; every kernel holds sixteen LDS-DMA requests: a slot's eight pieces in front of the K loop and eight
per slot inside it (four per load interval of the slot's first step).      python scripts/check_gemm_isa.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "eddie-wang-hackathon2023_amd", "csrc", "gemm_f16p.hip")
out = os.path.join(tempfile.mkdtemp(), "gemm_f16p.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
bad = 0
seen = 0
for m in re.finditer(r"^(_ZN2wm16gemm_f16p_kernelILi(\d)ELb(\d)EEEvNS_13GemmBigParamsE):.*?s_endpgm", text, re.S | re.M):
    name, act, simple, body = m.group(1), int(m.group(2)), m.group(3) == "1", m.group(0)
    x4, x2 = len(re.findall(r"global_store_dwordx4\s", body)), len(re.findall(r"global_store_dwordx2\s", body))
    ok = (x2 == 0 and x4 == (64 if act == 0 else 16)) if simple else (x4 == 0 and x2 > 0 and x2 % 32 == 0)
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", text[text.index(".amdhsa_kernel " + name):]).group(1))
    if simple or act < 2: ok = ok and scratch == 0
    dma = len(re.findall(r"global_load_lds_dwordx4\s", body))          # 8 pieces in front of the loop + 8 per slot inside it
    ok = ok and dma == 16
    print(("ok   " if ok else "BAD  ") + name, f"16-byte stores {x4}, 8-byte stores {x2}, scratch {scratch} B, LDS-DMA requests {dma}")
    bad += not ok
    seen += 1
if seen != 6:
    print(f"BAD  expected 6 kernels, found {seen}")
    bad += 1
sys.exit(1 if bad else 0)
