"""Build-time check of gemm_rows.hip's generated code (no GPU needed): in every variant's prologue all LDS-DMA requests
(global_load_lds) precede the first weight load (global_load_dwordx4), and exactly RING (10; 8 in the 8-wave variants) weight loads sit
between the last DMA request and the counted `s_waitcnt vmcnt(RING)` that stands for "this wave's input rows have landed".
    python scripts/check_rows_isa.py   (runs hipcc -S on csrc/gemm_rows.hip)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "eddie-wang-hackathon2023_amd", "csrc", "gemm_rows.hip")
out = os.path.join(tempfile.mkdtemp(), "gemm_rows.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
bad = 0
for m in re.finditer(r"^(_ZN2wm16gemm_rows_kernel\w+):.*?s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(0).splitlines()
    ring = 8 if name.endswith("ELi8EEEvNS_15GemvSmallParamsEii") else 10
    wait = next((i for i, l in enumerate(body) if f"s_waitcnt vmcnt({ring})" in l and "lgkmcnt" not in l), None)
    dma = [i for i, l in enumerate(body) if "global_load_lds" in l and (wait is None or i < wait)]
    wl = [i for i, l in enumerate(body) if re.search(r"global_load_dwordx4\s", l) and (wait is None or i < wait)]
    ok = wait is not None and dma and wl and min(wl) > max(dma) and len(wl) == ring
    print(("ok   " if ok else "BAD  ") + name, f"dma={len(dma)} weight_loads_before_wait={len(wl)}")
    bad += not ok
sys.exit(1 if bad else 0)
