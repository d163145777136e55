// Weight-streaming GEMM for the decode path: M <= 128 activation rows (M = B for a decode step,
// 3B for the prefill) against a weight matrix that is read from HBM exactly once.
//
// Replaces WeightOnlyQuantMatmulPlugin::enqueue's M == 1 GEMV
// (R/cpp/tensorrt_llm/kernels/weightOnlyMatrixVectorMultiplication.cu:136-205,371-378) and its
// small-M CUTLASS branch (weightOnlyQuantMatmulPlugin.cpp:182-197), plus the fp16 MatMul of the
// non-quantised engines and the tied logits projection (whisper/model.py:290).
//
// Weight layout ("tile-linear", written by weight.py): the [N][K] matrix is cut into tiles of
// 16 output channels x KT input channels, KT = 64 (int8) or 32 (fp16), each tile 1024 contiguous
// bytes ordered by lane: lane l holds 16 bytes = channel (l & 15), inputs [KT/4 * (l >> 4), +KT/4).
// Tiles are ordered [n_block][k_tile].  One wave-wide 16-byte load therefore fetches 1 KiB of
// contiguous HBM and lands every lane's bytes already in the B-operand layout of
// v_mfma_f32_16x16x32_f16 (B[k = 8 * (lane >> 4) + j][n = lane & 15]); int8 tiles feed two MFMAs
// (their first / second 8 inputs per lane) after an exact int8 -> fp16 expansion in registers.
// Packed int4 (--weight_only_precision int4; weightOnlyMatrixVectorMultiplication.cu:207-277 in the
// reference): KT = 128, a lane's 16 bytes are 32 biased nibbles (q + 8) for inputs 32 * (l >> 4) .. +32 of
// channel l & 15, i.e. four MFMA B operands; inside each 32-bit word the nibble of input j sits at position
// j/2 (j even) or 4 + j/2 (j odd), so that (word >> 4s) & 0x000F000F yields the fp16 pair (input 2s, 2s+1)
// after OR-ing in the exponent of 1024 and subtracting 1032.
// There is no LDS round trip for the streamed operand; only the small activation block is staged
// in LDS (shared by the 4 waves of a workgroup, each wave owning a different 16-channel block).
//
// Parallelism: grid = ceil(n_blocks / 4) x ksplit workgroups.  K slices write fp32 partial slabs
// part[s][m][n] (already multiplied by the per-channel scale); the consumer kernel sums the slabs
// in a fixed order (deterministic, unlike atomics) together with bias / residual / LN / GELU.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace wm {


template <int WB, int MT, int NW>       // WB: weight bits (16, 8, 4); NW waves per workgroup share one staged activation chunk
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmSkinnyParams p) {
    // Issue priority over the waves of the other utterance groups' K/V streams that share the SIMD: those keep the
    // vector-memory path busy for ~200 us at a time, this kernel is one link of a chain of dependent launches.  Next to
    // the stream the chain takes 355 instead of 473 us per layer (B = 576, three groups: 24.6 instead of 25.7 ms per step).
    __builtin_amdgcn_s_setprio(3);

    constexpr int KC = MT > 4 ? 128 : 256;    // activation chunk staged per barrier (inputs); LDS <= 35 KB
    constexpr int A_ROW = KC * 2 + 16;        // LDS row stride in bytes
    constexpr bool W8 = WB == 8;
    constexpr int KT = WB == 4 ? 128 : (WB == 8 ? 64 : 32);   // inputs per 1 KiB weight tile
    constexpr int TPC = KC / KT;              // tiles per chunk
    __shared__ __attribute__((aligned(16))) unsigned char sA[MT * 16 * A_ROW];

    // blockIdx.y: row split -- rows [MT * 16 * y, +MT * 16) of the launch (slab mode at many rows: a workgroup then stages a
    // third of the activation slice; the K = 5120 Linear at 192 rows ran on 40 workgroups that each moved 480 KB of
    // activations through LDS, 30 us alone and 62 us next to the other groups' streams -- the longest link of the chain).
    // Rows are independent: the split changes nothing in a row's arithmetic.
    const int row0 = blockIdx.y * (MT * 16);
    p.A += (size_t)row0 * p.lda;
    if (p.part) p.part += (size_t)row0 * (p.n_blocks * 16);
    if (p.out) p.out += (size_t)row0 * p.ldc;
    if (p.part_sstride == 0) p.part_sstride = (long)p.M * p.n_blocks * 16;
    p.M = min(p.M - row0, MT * 16);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nwg_n = (p.n_blocks + NW - 1) / NW;
    const int bn = blockIdx.x % nwg_n, ks = blockIdx.x / nwg_n;
    const int nb = bn * NW + wid;                          // this wave's 16-channel block
    const bool wave_active = nb < p.n_blocks;
    // the per-channel scale of the epilogue: an unconditional load, requested now (read behind the K loop -- and behind a test, which
    // makes hipcc wait for it inside the branch -- it was a round trip of its own at the end of every launch)
    const bool has_scale = WB != 16 && p.scale != nullptr;
    const h16 sc_raw = *((has_scale ? p.scale : (const h16*)p.Wt) + (has_scale && wave_active ? nb * 16 + (lane & 15) : 0));

    const int kt_total = p.K / KT;
    const int tps = (kt_total + p.ksplit - 1) / p.ksplit;  // tiles per split
    const int t_begin = ks * tps;
    const int t_end = min(kt_total, t_begin + tps);

    const u32x4* wt = (const u32x4*)p.Wt + ((size_t)(wave_active ? nb : 0) * kt_total) * 64 + lane;

    float4v acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};

    u32x4 wreg[TPC];
    auto load_w = [&](int t0) {
#pragma unroll
        for (int i = 0; i < TPC; ++i) {
            const int t = t0 + i;
            if (wave_active && t < t_end) wreg[i] = __builtin_nontemporal_load(wt + (size_t)t * 64);
        }
    };
    // The activation chunk goes global -> registers -> LDS, and the registers of chunk c + 1 are requested before the
    // MFMAs of chunk c: a workgroup then waits for memory once (weights and first chunk together) instead of once per
    // chunk.  Next to a K/V stream that keeps the HBM queues full a round trip costs ~3x its unloaded time, and these
    // kernels are nothing but a few round trips (scripts/contention_probe.py).
    constexpr int A_VEC = MT * 16 * (KC / 8);             // 16-byte pieces of a chunk
    constexpr int A_PER_T = (A_VEC + NW * 64 - 1) / (NW * 64);
    uint4 areg[A_PER_T];
    auto load_a = [&](int t0) {
        const int k0 = t0 * KT;
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j) {
            const int c = tid + j * (NW * 64);
            const int r = c / (KC / 8), cc = c % (KC / 8);
            uint4 v = make_uint4(0, 0, 0, 0);              // zero rows >= M, zero columns >= K
            if (c < A_VEC && r < p.M && k0 + cc * 8 < p.K) v = *(const uint4*)(p.A + (size_t)r * p.lda + k0 + cc * 8);
            areg[j] = v;
        }
    };
    load_w(t_begin);
    load_a(t_begin);

    const int frag_row = (lane & 15) * A_ROW;
    for (int t0 = t_begin; t0 < t_end; t0 += TPC) {
        // ---- A[:, t0*KT .. +KC) into LDS -----------------------------------------------------
#pragma unroll
        for (int j = 0; j < A_PER_T; ++j) {
            const int c = tid + j * (NW * 64);
            if (c < A_VEC) *(uint4*)(sA + (c / (KC / 8)) * A_ROW + (c % (KC / 8)) * 16) = areg[j];
        }
        __syncthreads();
        u32x4 wcur[TPC];
#pragma unroll
        for (int i = 0; i < TPC; ++i) wcur[i] = wreg[i];
        if (t0 + TPC < t_end) {                       // next chunk's weights and activations fly during the MFMAs
            load_w(t0 + TPC);
            load_a(t0 + TPC);
        }
        if (wave_active) {
#pragma unroll
            for (int i = 0; i < TPC; ++i) {
                if (t0 + i < t_end) {
                    if (WB == 4) {
                        // lane's 32 inputs start at 32 * (lane >> 4) inside the 128-wide tile; word m feeds MFMA m
                        const unsigned char* ab = sA + frag_row + (i * 128 + (lane >> 4) * 32) * 2;
                        const uint32_t wv[4] = {wcur[i].x, wcur[i].y, wcur[i].z, wcur[i].w};
                        const half2v bias8 = {(h16)1032.0f, (h16)1032.0f};
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            half8v b;
#pragma unroll
                            for (int sft = 0; sft < 4; ++sft) {
                                const uint32_t bits = ((wv[m] >> (4 * sft)) & 0x000F000Fu) | 0x64006400u;   // fp16 (1024 + u) x 2
                                const half2v pr = __builtin_bit_cast(half2v, bits) - bias8;                  // u - 8 = q, exact
                                b[2 * sft] = pr[0]; b[2 * sft + 1] = pr[1];
                            }
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                const half8v a = *(const half8v*)(ab + mt * 16 * A_ROW + m * 16);
                                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[mt], 0, 0, 0);
                            }
                        }
                    } else if (W8) {
                        half2v h[8];
                        cvt_s8x4_f16x4(wcur[i].x, h[0], h[1]);
                        cvt_s8x4_f16x4(wcur[i].y, h[2], h[3]);
                        cvt_s8x4_f16x4(wcur[i].z, h[4], h[5]);
                        cvt_s8x4_f16x4(wcur[i].w, h[6], h[7]);
                        half8v b0, b1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            b0[2 * j] = h[j][0]; b0[2 * j + 1] = h[j][1];
                            b1[2 * j] = h[4 + j][0]; b1[2 * j + 1] = h[4 + j][1];
                        }
                        // lane's 16 inputs start at 16 * (lane >> 4) inside the 64-wide tile
                        const unsigned char* ab = sA + frag_row + (i * 64 + (lane >> 4) * 16) * 2;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const half8v a0 = *(const half8v*)(ab + mt * 16 * A_ROW);
                            const half8v a1 = *(const half8v*)(ab + mt * 16 * A_ROW + 16);
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[mt], 0, 0, 0);
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[mt], 0, 0, 0);
                        }
                    } else {
                        const half8v b0 = __builtin_bit_cast(half8v, wcur[i]);
                        const unsigned char* ab = sA + frag_row + (i * 32 + (lane >> 4) * 8) * 2;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const half8v a0 = *(const half8v*)(ab + mt * 16 * A_ROW);
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[mt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    if (!wave_active) return;
    const int col = nb * 16 + (lane & 15);
    const float sc = has_scale ? (float)sc_raw : 1.0f;
    const int ldp = p.n_blocks * 16;
    const size_t sstride = p.part_sstride ? (size_t)p.part_sstride : (size_t)p.M * ldp;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + (lane >> 4) * 4 + r;
            if (row >= p.M) continue;
            const float v = acc[mt][r] * sc;
            if (p.out) {
                if (col < p.n_valid) p.out[(size_t)row * p.ldc + col] = (h16)v;
            } else {
                p.part[(size_t)ks * sstride + (size_t)row * ldp + col] = v;
            }
        }
    }
}


static inline int skinny_kt(int w8) { return w8 == 4 ? 128 : (w8 ? 64 : 32); }    // w8: 0 fp16, 1 int8, 4 packed int4

int skinny_default_ksplit(int M, int K, int n_blocks, int w8) {
    // The split depends on the weight matrix only, never on the number of rows: the order in which a row's K slices
    // are summed (by the consumer kernels) must not change with the batch the row is in.
    (void)M;
    const int kt_total = K / skinny_kt(w8);
    const int nwg_n = (n_blocks + 7) / 8;
    int s = (384 + nwg_n - 1) / nwg_n;                 // aim for ~3000 waves in flight (12 per CU)
    const int min_tiles = w8 == 4 ? 1 : (w8 ? 2 : 4);  // at least 128 inputs per slice
    s = min(s, max(1, kt_total / min_tiles));
    // the fp32 slabs (s * rows * N * 4 B) are written and re-read through L2 / Infinity Cache: keep them under
    // ~16 MB at the largest row count a launch takes, so that they stay on-die
    const long slab = 128L * n_blocks * 16 * 4;          // sized for 128 rows whatever the launch takes: the split must not depend on M
    s = min(s, (int)max(1L, (16L << 20) / slab));
    // every slab is written and re-read (by the row kernel) as fp32: with 8 slices that is 115 MB per layer for 128 rows,
    // 12 % of what the cross-attention kernel streams -- and next to that kernel the chain of short kernels is
    // throughput-bound, not latency-bound.  4 slices: 26.0 instead of 26.9 ms per decode step at B = 576 (8 are 5 % faster
    // for a kernel running alone, scripts/bench_skinny.py).  WM_KSPLIT_CAP overrides (experiments).
    static const int cap_env = lab_env_int("WM_KSPLIT_CAP", 4);
    s = min(s, cap_env);
    return max(1, s);
}

// Waves per workgroup for the multi-tile shapes (3 or more MFMA row tiles).  8 waves share one staged activation chunk
// (half the L2 traffic for A), but an 8-wave workgroup needs two waves' registers on every SIMD (2 x 136 at 12 row tiles):
// it cannot be dispatched to a CU on which three workgroups of another group's K/V stream (3 x 96 registers per SIMD) are
// resident -- which is every CU once the encoder of the next batch holds part of the chip -- and then the groups' chains
// no longer overlap the other groups' streams.  4-wave workgroups (one wave per SIMD) fit.  The arithmetic does not depend
// on the choice (same K slices, same order).  WM_SKINNY_NW=4|8 overrides.
static int skinny_nw() {
    static const int nw = lab_env_int("WM_SKINNY_NW", 8) == 4 ? 4 : 8;
    return nw;
}

// Row tiles per workgroup above 64 rows in slab mode (WM_SKINNY_MT=4|6|8|12|16 overrides; 16 = rounds 1-2: no row split)
static int skinny_split_mt() {
    static const int mt = [] { const int x = lab_env_int("WM_SKINNY_MT", 4); return (x == 4 || x == 6 || x == 8 || x == 12 || x == 16) ? x : 4; }();
    return mt;
}

template <int WB>
static int launch_mt(const GemmSkinnyParams& p, hipStream_t stream) {
    int mt = (p.M + 15) / 16;
    int n_ms = 1;
    if (p.part && !p.out && mt > skinny_split_mt()) { n_ms = (mt + skinny_split_mt() - 1) / skinny_split_mt(); mt = skinny_split_mt(); }
    const dim3 g4(((p.n_blocks + 3) / 4) * p.ksplit, n_ms), g8(((p.n_blocks + 7) / 8) * p.ksplit, n_ms);
    if (skinny_nw() == 4 && mt >= 3) {
        switch (mt) {
            case 3: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 3, 4>), g4, dim3(256), 0, stream, p); break;
            case 4: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 4, 4>), g4, dim3(256), 0, stream, p); break;
            case 5: case 6: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 6, 4>), g4, dim3(256), 0, stream, p); break;
            case 7: case 8: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 8, 4>), g4, dim3(256), 0, stream, p); break;
            case 9: case 10: case 11: case 12: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 12, 4>), g4, dim3(256), 0, stream, p); break;
            default: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 16, 4>), g4, dim3(256), 0, stream, p); break;
        }
        return 0;
    }
    switch (mt) {
        case 1: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 1, 4>), g4, dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 2, 4>), g4, dim3(256), 0, stream, p); break;
        case 3: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 3, 8>), g8, dim3(512), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 4, 8>), g8, dim3(512), 0, stream, p); break;
        case 5: case 6: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 6, 8>), g8, dim3(512), 0, stream, p); break;
        case 7: case 8: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 8, 8>), g8, dim3(512), 0, stream, p); break;
        case 9: case 10: case 11: case 12: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 12, 8>), g8, dim3(512), 0, stream, p); break;
        default: hipLaunchKernelGGL((gemm_skinny_kernel<WB, 16, 8>), g8, dim3(512), 0, stream, p); break;
    }
    return 0;
}

int launch_gemm_skinny(const GemmSkinnyParams& p, hipStream_t stream) {
    WM_REQUIRE(p.M >= 1 && p.M <= SKINNY_MAX_M, "gemm_skinny: M=%d out of range [1,%d]", p.M, SKINNY_MAX_M);
    const int KT = skinny_kt(p.w8);
    WM_REQUIRE(p.w8 == 0 || p.w8 == 1 || p.w8 == 4, "gemm_skinny: w8=%d (0 fp16, 1 int8, 4 packed int4)", p.w8);
    WM_REQUIRE(p.K % KT == 0, "gemm_skinny: K=%d must be a multiple of %d", p.K, KT);
    WM_REQUIRE(p.lda % 8 == 0, "gemm_skinny: lda=%d must be a multiple of 8", p.lda);
    WM_REQUIRE(p.ksplit >= 1, "gemm_skinny: ksplit must be >= 1");
    WM_REQUIRE(p.out == nullptr || p.ksplit == 1, "gemm_skinny: direct output needs ksplit == 1");
    WM_REQUIRE(p.out != nullptr || p.part != nullptr, "gemm_skinny: no output buffer");
    if (p.w8 == 4) launch_mt<4>(p, stream); else if (p.w8) launch_mt<8>(p, stream); else launch_mt<16>(p, stream);
    WM_LAUNCH_CHECK(stream, "gemm_skinny");
    return 0;
}

}  // namespace wm
