"""Lab: where a batch-1 token step spends its time INSIDE the one-launch decoder (csrc/gemv_chain.hip).
    python scripts/lab/chain_stamps.py build            # writes build/lab/gemv_chain_stamps.hip and builds build/lab/libwm_stamps<wg>.so (wg = 0, 160, 200)
    WM_LIBRARY_PATH=build/lab/libwm_stamps200.so python scripts/lab/chain_stamps.py read     # on the GPU box
A copy of the kernel in which ONE thread of ONE workgroup (0: owns output groups of every Linear; 160: a self-attention head; 200: a
cross-attention (head, piece)) writes the 100 MHz device clock at the stage boundaries of every layer; the reader averages over layers
1 .. L-2 of the last decode step.  Stamping costs ~ 5 % of the step.  Nothing here ships."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(coarse=False):
    """coarse: the stage-boundary stamps and the Linear stages' inner ones only (the 3-8-row kernels' attention stages have other markers);
    builds build/lab/libwm_cstamps<wg>.so for wg = 0 and 250"""
    s = open(os.path.join(ROOT, "eddie-wang-hackathon2023_amd/csrc/gemv_chain.hip")).read()

    def ins_before(s, marker, text):
        assert s.count(marker) == 1, (marker, s.count(marker))
        return s.replace(marker, text + marker, 1)

    def ins_after(s, marker, text):
        assert s.count(marker) == 1, (marker, s.count(marker))
        return s.replace(marker, marker + text, 1)
    s = s.replace('''// Where a stage without LayerNorm takes its input row from (wave-uniform):''', '''__device__ unsigned long long g_stamps[64 * 32];
__device__ unsigned long long g_stamps2[64 * 8 * 8];
// The stamps of a layer are collected in LDS and written out once, at the start of the next layer: a stamp that went to memory
// directly would be waited for (its acknowledgement, ~ 2 us) at the next workgroup barrier and measure mostly itself.
__shared__ unsigned long long s_stamps[32 + 64];
__shared__ int s_stamp_layer;
__shared__ int s_cur_stage;
__shared__ unsigned long long s_arrive[8];
extern "C" int wm_lab_chain_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
extern "C" int wm_lab_chain_stamps2(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps2), sizeof(g_stamps2)); }
#define STAMP_ON (threadIdx.x == 0 && blockIdx.x == WM_STAMP_WG)
#define STAMP(k) do { if (STAMP_ON) s_stamps[k] = wall_clock64(); } while (0)
#define STAMP2(k) do { if (STAMP_ON) s_stamps[32 + s_cur_stage * 8 + (k)] = wall_clock64(); } while (0)
#define STAMP_FLUSH(layer) do { if (STAMP_ON) { const int pl = s_stamp_layer; if (pl >= 0 && pl < 64) { for (int i_ = 0; i_ < 32; ++i_) g_stamps[pl * 32 + i_] = s_stamps[i_]; \\
    for (int i_ = 0; i_ < 64; ++i_) g_stamps2[pl * 64 + i_] = s_stamps[32 + i_]; } for (int i_ = 0; i_ < 96; ++i_) s_stamps[i_] = 0; s_stamp_layer = (layer); } } while (0)
// Where a stage without LayerNorm takes its input row from (wave-uniform):''', 1)
    s = ins_before(s, '    // ---- 1. every weight tile', '    STAMP2(0);\n')
    s = ins_after(s, '        ok = sweep_granules16<2 * XP>(gx, first, tag, val, p.err, lane);\n', '        STAMP2(1);\n')
    s = ins_after(s, '        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // this wave\'s LDS writes before its reads\n', '        STAMP2(1);\n')
    s = ins_before(s, '    // ---- 3. multiply', '    STAMP2(2);\n')
    s = ins_before(s, '    // ---- 4. epilogue', '    STAMP2(3);\n')
    s = ins_before(s, '    if (st.mode == 2) own_valid = true;\n    __syncthreads();                                          // s_red / s_in', '    STAMP2(4);\n')
    s = ins_after(s, '        const unsigned epoch0 = gen | ((unsigned)(whole ? (l & 63) : p.launch_id) << 3);\n',
                  '        STAMP_FLUSH(whole ? l : p.launch_id);\n        STAMP(0);\n')
    s = ins_before(s, '        const int s_first = l < 0 ? 5 : 0;', '        STAMP(7);\n')
    s = ins_before(s, '    bool own_valid = false, x_in_granules = false;\n    for (int l = whole ? -1 : 0;', '    if (STAMP_ON) s_stamp_layer = -1;\n')
    s = ins_after(s, '            if (st.mode == 2) x_in_granules = true;               // the residual row of the stages behind: this launch\'s granules\n', '            STAMP(1 + s);\n')
    s = ins_before(s, '                // the NEXT layer\'s K / V rows set out now', '                STAMP(8);\n')
    s = ins_before(s, '            if (wide) chain_stage<WB, true, false, NR>(p, st, s, epoch, own_valid', '            if (STAMP_ON) s_cur_stage = s;\n')
    s = ins_after(s, '        chain_merge_tagged<NR>(p, tag, s_in);\n', '        STAMP(9);\n')           # (the merge runs inside the stage since round 5)
    if coarse:
        s = s.replace('constexpr size_t CHAIN_DYN_LDS = 100 * 1024;', 'constexpr size_t CHAIN_DYN_LDS = 97 * 1024;      // (lab: room for the stamps)', 1)
        s = s.replace('constexpr size_t CHAIN_DYN_LDS8 = 24 * 1024;', 'constexpr size_t CHAIN_DYN_LDS8 = 22 * 1024;      // (lab: room for the stamps)', 1)
        out = os.path.join(ROOT, "build/lab")
        os.makedirs(out, exist_ok=True)
        open(os.path.join(out, "gemv_chain_cstamps.hip"), "w").write(s)
        for wg in (0, 250):
            env = dict(os.environ, SRC=os.path.join(out, "gemv_chain_cstamps.hip"))
            r = subprocess.run([os.path.join(ROOT, "scripts/lab/build_variant.sh"), f"cstamps{wg}", "gemv_chain.hip", f"-DWM_STAMP_WG={wg}"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            print(r.stdout.strip().splitlines()[-1])
        return
    a = s.index('__device__ __forceinline__ void chain_cross_stage(')
    b = s.index('__device__ __forceinline__ void chain_merge_tagged(', a)
    body = s[a:b]
    for tag, k in (('(A) rows and q sums are in LDS', 11), ('(B)', 12), ('(C)', 13), ('(D)', 14)):
        line = [l for l in body.splitlines(True) if l.strip().startswith('__syncthreads();') and l.rstrip().endswith('// ' + tag)]
        assert len(line) == 1, tag
        body = body.replace(line[0], ('    STAMP(10);\n' if k == 11 else '') + line[0] + f'    STAMP({k});\n', 1)
    # finer: inside P.V (15: rows and probabilities read, products added; 23: the cross-lane sums done)
    m1 = '#pragma unroll\n        for (int e = 0; e < DPL; ++e) o[e] += wave_dpp<0x128>(o[e]);'
    assert body.count(m1) == 1
    body = body.replace(m1, '        STAMP(15);\n' + m1, 1)
    m2 = '        if (rowi == 0) {\n#pragma unroll\n            for (int e = 0; e < DPL; ++e) s_o[wid][sub * DPL + e] = o[e];'
    assert body.count(m2) == 1
    body = body.replace(m2, '        STAMP(23);\n' + m2, 1)
    s = s[:a] + body + s[b:]
    a, b = s.index('__device__ __forceinline__ void chain_self_stage('), s.index('template <int WB, bool I8KV, int NR>\n__global__ __launch_bounds__(512) void gemv_chain_kernel')
    body = s[a:b].replace('        q = r16(q + (la.self_bias ? (float)bq_raw : 0.f));', '        STAMP(17);\n        q = r16(q + (la.self_bias ? (float)bq_raw : 0.f));', 1)
    assert 'STAMP(17)' in body
    parts = body.split('    __syncthreads();\n')
    assert len(parts) == 7, len(parts)
    body = (parts[0] + '    STAMP(18);\n    if (blockIdx.x == WM_STAMP_WG && lane == 0) s_arrive[wid] = wall_clock64();\n    __syncthreads();\n    STAMP(19);\n    if (STAMP_ON) for (int w_ = 0; w_ < 8; ++w_) s_stamps[24 + w_] = s_arrive[w_];\n' + parts[1] + '    STAMP(15);\n    __syncthreads();\n' + parts[2] + '    __syncthreads();\n    STAMP(23);\n' + parts[3]
            + '    __syncthreads();\n    STAMP(20);\n' + parts[4] + '    __syncthreads();\n    STAMP(21);\n' + parts[5] + '    STAMP(22);\n    __syncthreads();\n' + parts[6])
    body = body.replace('    const float t_dq = la.self_kv_scale;', '    STAMP(16);\n    const float t_dq = la.self_kv_scale;', 1)
    s = s[:a] + body + s[b:]
    s = s.replace('constexpr size_t CHAIN_DYN_LDS = 100 * 1024;', 'constexpr size_t CHAIN_DYN_LDS = 97 * 1024;      // (lab: room for the stamps)', 1)
    out = os.path.join(ROOT, "build/lab")
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, "gemv_chain_stamps.hip"), "w").write(s)
    for wg in (0, 160, 200, 60, 120):       # 0: owner of output groups; one row: 160 self-attention, 200 cross-attention; two rows: 60 self-, 120 cross-attention
        env = dict(os.environ, SRC=os.path.join(out, "gemv_chain_stamps.hip"))
        r = subprocess.run([os.path.join(ROOT, "scripts/lab/build_variant.sh"), f"stamps{wg}", "gemv_chain.hip", f"-DWM_STAMP_WG={wg}"], env=env, stdout=subprocess.PIPE, text=True)
        print(r.stdout.strip().splitlines()[-1])


def read():
    import ctypes as C
    from pathlib import Path
    sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd"), os.path.join(ROOT, "tests")]
    import numpy as np
    import torch
    import native, synthetic, bench
    from decoding import WhisperDecoding
    from encoding import WhisperEncoding
    from types import SimpleNamespace
    from synthetic import synthetic_mel
    lib = native.load_library()
    model = "large-v2"
    n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dims = SimpleNamespace(**synthetic.DIMS[model])
    sys.argv = ["bench.py", "--model", model]
    args = bench.parse()
    eng_dir = Path(args.engine_cache) / f"{args.model}-{args.config}-seed{args.seed}"
    if not (eng_dir / "decoder_config.json").exists():
        eng_dir.parent.mkdir(parents=True, exist_ok=True)
        bench.build_engines(args, eng_dir)
    enc, dec = WhisperEncoding(eng_dir), WhisperDecoding(eng_dir)
    dec.sample_len = 40
    xa = enc.get_audio_features(synthetic_mel(n_rows, 2 * dims.n_audio_ctx, dims.n_mels, 81).cuda())
    dec.detect_language(xa)
    for _ in range(3):
        dec.main_loop(xa, ignore_eot=True)
    torch.cuda.synchronize()
    st = np.zeros(64 * 32, dtype=np.uint64)
    lib.wm_lab_chain_stamps.argtypes = [C.c_void_p]
    lib.wm_lab_chain_stamps(st.ctypes.data)
    st = st.reshape(64, 32).astype(np.int64)
    L = dims.n_text_layer
    x = np.array([st[i] for i in range(1, L - 2)], dtype=np.float64) / 100.0
    nx = np.array([st[i + 1][0] for i in range(1, L - 2)], dtype=np.float64) / 100.0
    cols = [("self-attention (from layer start)", 7, 0), ("out", 1, 7), ("cq", 2, 1), ("cross-attention", 8, 2), ("merge of the pieces", 9, 8), ("cout", 3, 9),
            ("mlp1", 4, 3), ("mlp2", 5, 4), ("qkv of the next layer", 6, 5)]
    print(f"library {os.environ.get('WM_LIBRARY_PATH')}, {n_rows} row(s): one workgroup's view, us per layer (mean over layers 1..{L - 3} of the last step at {dec.sample_len} tokens)")
    tot = 0.0
    for name, hi, lo in cols:
        d = x[:, hi] - x[:, lo]
        print(f"  {name:36s} {d.mean():6.2f}  (min {d.min():.2f} max {d.max():.2f})")
        tot += d.mean()
    tot += (nx - x[:, 6]).mean()
    print(f"  per layer {tot:.2f} us; x {L} = {tot * L / 1000:.3f} ms")
    s2 = np.zeros(64 * 8 * 8, dtype=np.uint64)
    lib.wm_lab_chain_stamps2.argtypes = [C.c_void_p]
    lib.wm_lab_chain_stamps2(s2.ctypes.data)
    s2 = s2.reshape(64, 8, 8).astype(np.int64)
    print("  Linear stages (wave 0 of slot 0): wait for the input | LayerNorm / fragments | multiply (weights waited for here) | epilogue + publish")
    for sidx, name in enumerate(["out", "cq", "cout", "mlp1", "mlp2", "qkv"]):
        y = np.array([s2[i, sidx] for i in range(1, L - 2)], dtype=np.float64) / 100.0
        if y[:, 0].min() <= 0:
            print(f"    {name:6s} (idle in this workgroup)")
            continue
        has_b = y[:, 1].min() > 0 and (y[:, 1] >= y[:, 0]).all()
        b = y[:, 1] if has_b else y[:, 0]
        print(f"    {name:6s} {np.mean(b - y[:, 0]):5.2f} | {np.mean(y[:, 2] - b):5.2f} | {np.mean(y[:, 3] - y[:, 2]):5.2f} | {np.mean(y[:, 4] - y[:, 3]):5.2f}")
    if st[1, 14] > st[1, 10] > 0:
        print("  cross-attention stage: cq end -> q swept %.2f | scores %.2f | max / exp / sum %.2f | P.V %.2f | sums + publish %.2f" % (
            np.mean(x[:, 11] - x[:, 2]), np.mean(x[:, 12] - x[:, 11]), np.mean(x[:, 13] - x[:, 12]), np.mean(x[:, 14] - x[:, 13]), np.mean(x[:, 8] - x[:, 14])))
        if st[1, 23] > st[1, 15] > 0:
            print("    inside P.V: barrier (C) -> rows read and products added %.2f | cross-lane sums %.2f | LDS write + barrier (D) %.2f" % (
                np.mean(x[:, 15] - x[:, 13]), np.mean(x[:, 23] - x[:, 15]), np.mean(x[:, 14] - x[:, 23])))
    if st[1, 22] > st[1, 16] > 0:
        print("  self-attention stage: table / address prologue %.2f | wait for the qkv sums %.2f | q, k, v formed %.2f | first barrier %.2f | scores + softmax (2 barriers) %.2f | P.V %.2f | sums + publish %.2f" % (
            np.mean(x[:, 16] - x[:, 0]), np.mean(x[:, 17] - x[:, 16]), np.mean(x[:, 18] - x[:, 17]), np.mean(x[:, 19] - x[:, 18]), np.mean(x[:, 20] - x[:, 19]),
            np.mean(x[:, 21] - x[:, 20]), np.mean(x[:, 22] - x[:, 21])))
        if st[1, 23] > st[1, 15] > 0:
            print("    inside scores + softmax: scores of wave 0 (to its arrival at barrier 2) %.2f | barrier 2 + exponentials + sum + barrier 3 %.2f | normalise + barrier 4 %.2f" % (
                np.mean(x[:, 15] - x[:, 19]), np.mean(x[:, 23] - x[:, 15]), np.mean(x[:, 20] - x[:, 23])))
        print("    arrival of waves 0..7 at the stage's first barrier, us after (negative: before) wave 0 starts the layer:", " ".join("%.2f" % np.mean(x[:, 24 + w] - x[:, 0]) for w in range(8)), "| released %.2f" % np.mean(x[:, 19] - x[:, 0]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("build", "build-coarse"):
        build(coarse=sys.argv[1] == "build-coarse")
    else:
        read()
