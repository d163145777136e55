"""Build-time budget of the encoder attention's main loop (no GPU needed): instruction counts per 64-key tile and wave in the
generated code of attn_encoder_kernel<4, *> (hipcc -S), against the floor the arithmetic contract sets (csrc/attn_encoder.hip:6-8:
scores rounded to fp16, softmax in fp32, probabilities rounded to fp16).

Per tile a wave owns 64 queries x 64 keys = 64 scores per lane.  The contract fixes, per score: one exponential (v_exp_f32, 8
issue cycles), one fused multiply-add on the fp16 score ((s - m) * log2 e: v_fma_mix_f32, 4), half a pair conversion each for the
score and the probability (v_cvt_pk_f16_f32, 4 per pair), half a v_dot2 for the row sum, and 56 / 64 of a maximum: ~ 22 issue
cycles per score = 1 400-1 800 per tile beside 64 MFMAs, which hold the SIMD's vector issue port for 8 of their 16 cycles (512).
The loop is therefore VALU-ISSUE bound at ~ 2 300 cycles per tile and wave against 1 024 cycles of matrix work (MFMA busy <= 44 %;
measured 32-34 %, 2 560 cycles: profiles/r3ad_pmc_stage_b192_pass1.txt, r3z_attn_encoder_tile_stamps.log): what is left for
scheduling is ~ 10 %.  This check keeps the loop AT that floor: a change that adds instructions per score, or puts a spill
inside the loop, fails here.       python scripts/check_attn_isa.py"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "eddie-wang-hackathon2023_amd", "csrc", "attn_encoder.hip")
out = os.path.join(tempfile.mkdtemp(), "attn_encoder.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
COST = collections.defaultdict(lambda: 4, {"v_exp_f32_e32": 8, "v_rcp_f32_e32": 8, "v_log_f32_e32": 8})
bad = 0
for persist in (0, 1):
    m = re.search(r"^_ZN2wm19attn_encoder_kernelILi4ELb%dEEEvNS_13AttnEncParamsE:.*?s_endpgm" % persist, text, re.S | re.M)
    body = m.group(0).split("\n")
    # the main tile loop: the innermost loop that holds 60+ MFMAs between its header label and its back branch
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        b = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if b and b.group(1) in labels and labels[b.group(1)] < i:
            seg = body[labels[b.group(1)]:i + 1]
            n_mfma = sum("v_mfma" in x for x in seg)
            if n_mfma >= 60 and (best is None or len(seg) < len(best)):
                best = seg
    assert best is not None, "main loop not found"
    ops = collections.Counter(l.split()[0] for l in best if l.strip() and not l.strip().startswith((";", ".")) and not re.match(r"^\S+:", l))
    valu = {k: v for k, v in ops.items() if k.startswith("v_") and "mfma" not in k}
    n_mfma = sum(v for k, v in ops.items() if "mfma" in k)
    issue = sum(COST[k] * v for k, v in valu.items()) + 8 * n_mfma
    spills = sum(v for k, v in ops.items() if k.startswith("scratch_"))
    # (round 6) the rescale factor exp2(m_old * log2 e - m_new * log2 e): two ROUNDED products and a subtraction, one per query block.  HIP's
    # __fmul_rn / __fsub_rn do not keep hipcc from fusing them into a v_fma_f32 (it did in a variant of this loop: 0.5 % of the outputs
    # one fp16 ulp off, profiles/r6v_*); the kernel pins both products with an empty asm, and the subtractions must show here
    unfused = ops["v_sub_f32_e32"] >= 4
    ok = (spills == 0 and 60 <= n_mfma <= 64 and ops["v_exp_f32_e32"] <= 68 and ops["v_cvt_pk_f16_f32"] <= 64 and ops["v_fma_mix_f32"] <= 64
          and sum(valu.values()) <= 420 and issue <= 2450 and unfused)
    print(("ok   " if ok else "BAD  ") + f"attn_encoder_kernel<4, {bool(persist)}> main loop: {n_mfma} MFMA, {sum(valu.values())} VALU "
          f"({ops['v_exp_f32_e32']} exp, {ops['v_cvt_pk_f16_f32']} cvt_pk, {ops['v_fma_mix_f32']} fma_mix, {ops['v_dot2c_f32_f16_e32']} dot2, "
          f"{ops['v_max3_f32'] + ops['v_max_f32_e32']} max, {ops['v_sub_f32_e32']} sub), {spills} scratch ops; issue cycles per tile and wave ~ {issue} "
          f"(MFMA work 1024)")
    bad += not ok
sys.exit(1 if bad else 0)
