// Weight-streaming Linear for decode groups of MORE than a handful of rows (round 3): the small-batch kernel's contract --
// one launch per Linear, LayerNorm of the input rows in the prologue, bias / GELU / residual in the epilogue, no fp32
// slabs in global memory, no row kernel (gemv_small.hip) -- cut the other way round.
//
//   [LayerNorm of the input rows]  ->  x . W^T  (tile-linear int8 / int4 / fp16 weights)  ->  bias / GELU / residual
//
// Replaces, like gemm_skinny.hip + rowops.hip's row kernel: WeightOnlyQuantMatmulPlugin::enqueue's small-M branch
// (weightOnlyQuantMatmulPlugin.cpp:182-197, weightOnlyMatrixVectorMultiplication.cu:136-277), the fp16 MatMul of the
// non-quantised engines, and the element-wise layers around them (LayerNorm normalization.py:6-30, bias add
// quantization/layer.py:311-312, gelu functional.py:2044-2056, residual adds whisper/model.py:61-122).
//
// Why.  At the bench's group size (192 rows) the split-K form spends its time on hand-overs, not on weights: every
// Linear writes 2-4 fp32 slabs (43 MB per layer and group, read back by the row kernel: 86 MB next to 23 MB of weights),
// runs on 40-160 workgroups that each stage a 192-row activation slice, and needs a second launch to finish its rows --
// twelve launches per layer, each a few dependent memory round trips that cost ~3x their unloaded time next to the other
// groups' K/V streams.  Here the ROWS are split over workgroups instead of K:
//   * a workgroup = 16 * MT rows x 64 output channels (4 waves, one 16-channel block each; 8 waves x 16 from 2560
//     channels on), the whole K per wave: the sums of a row never leave the accumulators, so the epilogue (bias, GELU,
//     residual add, fp16 rounding points as epilogue.h) runs in the same launch; the residual values of the in-place
//     mode are requested before the K loop;
//   * its input rows (16 * MT x K fp16) go global -> LDS by DMA (global_load_lds, 1 KiB per wave instruction, no
//     registers) into rows of whole KiB + 16 bytes (the MFMA A-fragment reads -- 16 rows x one 16-byte chunk -- are then
//     conflict-free); with LayerNorm every wave normalises the rows it requested in place (statistics by wave
//     reductions, the affine step, one LDS write: gemv_small's arithmetic, two rows per trip) while its weights are in
//     flight.  The kernel stays under 128 registers: a 4-wave workgroup then fits on a CU beside four waves per SIMD of
//     the other groups' K/V streams (4 x 96 + 128 = 512), which an 8-wave split-K workgroup (2 x 136) does not.  Every
//     channel group repeats the LayerNorm of its rows (~300 VALU instructions per row for one wave): the engine uses the
//     prologue for the projections behind an in-place residual add (cq, mlp1) and feeds qkv the rows the row kernel
//     behind mlp2 has normalised anyway;
//   * weights go HBM / L2 -> VGPRs in MFMA B-operand order (one wave-wide 16-byte load = one 1 KiB tile), a ring of
//     R tiles in flight per wave; the row splits of one channel group are given equal blockIdx % 8, i.e. one XCD under
//     round-robin placement (speed only), so the group's weights leave HBM once and are re-read from that XCD's L2.
// A row's sums are one fp32 accumulation chain over K in tile order, whatever the batch or the row split it is in: results
// do not depend on the launch shape.  (They are not the split-K path's sums bit for bit -- that path adds 2-4 partial sums.)
// K <= 1536 (a lane holds three 16-byte pieces of a row; the input block, 16 * MT rows of ceil(K / 512) KiB + 16 bytes, must fit the
// LDS): the projections that read the residual stream or the attention output (K = n_state); the MLP's second Linear
// (K = 4 n_state) stays on gemm_skinny.hip (K slices and row splits over workgroups) + the row kernel.
// Measured (MI355X, large-v2 int8, three groups of 192 rows, in situ per launch, profiles/r3k_chain_rows.log against
// r3j_chain_splitk.log): out / cout 23-29 us against 35-39 + 22-24 (GEMM + row kernel), cq 30 against 28 (+ the row kernel
// above), mlp1 66 against 46 + 28, qkv 38 against 41; token step 25.2 -> 23.4-23.7 ms.  Below ~40 rows per group the split-K
// pair is the quicker link (profiles/r3k_batch_sweep.txt): the engine switches there (engine.hip: rows_path_min_rows).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"
#include "epilogue.h"
#include "kernels.h"

namespace wm {

namespace rows {
constexpr int ring(int nw) { return nw == 8 ? 8 : 10; }     // weight tiles (1 KiB per wave) in flight per wave
constexpr int XP = 3;                   // 16-byte pieces of a row per lane (K <= 1536)
}  // namespace rows

// WB: weight bits (16, 8, 4); MT: 16-row MFMA tiles per workgroup; LN: LayerNorm of the input rows; NW: waves per workgroup = 16-channel
// blocks per workgroup (4, or 8 for wide outputs: half the workgroups -- one round over the chip at N = 5120 -- and half as many
// copies of the input block and of its LayerNorm, which every channel group redoes)
template <int WB, int MT, bool LN, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 5 : 4) void gemm_rows_kernel(GemvSmallParams p, int n_cg, int n_ms) {
    using namespace rows;
    constexpr int KT = WB == 4 ? 128 : (WB == 8 ? 64 : 32);   // inputs per 1 KiB weight tile
    constexpr int NM = KT / 32;                               // MFMAs (32-deep) per tile
    constexpr int ROWS = 16 * MT, RB = ROWS / NW;             // rows per workgroup, rows per wave in the prologue
    constexpr int RING = ring(NW);
    extern __shared__ __attribute__((aligned(1024))) unsigned char s_x[];      // [ROWS] rows of ceil(K / 512) KiB + 16 bytes

    // chain kernel next to the other groups' K/V streams (as gemm_skinny.hip)
    __builtin_amdgcn_s_setprio(3);

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, rl = lane & 15;
    // workgroup -> (channel group, row split): the row splits of a channel group share blockIdx % 8
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int cg = (j / n_ms) * 8 + xcd, ms = j - (j / n_ms) * n_ms;
    if (cg >= n_cg) return;
    const int nb = cg * NW + wid;
    const bool wave_active = nb < p.n_blocks;
    const int row0 = ms * ROWS;
    const int kt_total = p.K / KT;
    const int pieces_per_row = p.K >> 3;
    const int np = (pieces_per_row + 63) >> 6;                // 1 KiB pieces per row (<= XP)
    const int row_bytes = np * 1024 + 16;                     // LDS row stride: whole pieces + 16 bytes, so that the 16 rows of a fragment read sit on 16 different 16-byte slots

    // ---- requests, in the order they are waited for ---------------------------------------------------------------------------
    // 1. the input rows, global -> LDS by DMA (no registers: the workgroup has to fit beside the K/V streams' waves): wave w
    //    owns rows w, w + 4, ...; a row is np wave-wide 16-byte loads (lanes past K re-read the row's last chunk into the
    //    unused tail of the LDS row; rows past M re-read row M - 1 and are never stored)
#pragma unroll
    for (int jr = 0; jr < RB; ++jr) {
        const int r = wid + jr * NW;
        const h16* row = p.A + (size_t)min(row0 + r, p.M - 1) * p.lda;
#pragma unroll
        for (int u = 0; u < XP; ++u)
            if (u < np)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + min(lane + 64 * u, pieces_per_row - 1) * 8),
                                                 (__attribute__((address_space(3))) void*)(s_x + r * row_bytes + u * 1024), 16, 0, 0);
    }
    // 2. gamma / beta, by DMA as well (every wave requests the same bytes into the same place: a wave only relies on its own
    //    requests having landed, no barrier in front of the LayerNorm);  3. the first RING weight tiles (HBM, or the XCD's L2
    //    behind another row split)
    unsigned char* s_g = s_x + ROWS * row_bytes;              // gamma: np KiB, then beta
    if constexpr (LN) {
#pragma unroll
        for (int u = 0; u < XP; ++u)
            if (u < np) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ln_g + min(lane + 64 * u, pieces_per_row - 1) * 8),
                                                 (__attribute__((address_space(3))) void*)(s_g + u * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.ln_b + min(lane + 64 * u, pieces_per_row - 1) * 8),
                                                 (__attribute__((address_space(3))) void*)(s_g + (np + u) * 1024), 16, 0, 0);
            }
    }
    const u32x4* wt = (const u32x4*)p.Wt + (size_t)(wave_active ? nb : 0) * kt_total * 64 + lane;
    u32x4 wreg[RING];
    // (the counted wait below relies on the order: every DMA request, THEN exactly RING weight loads -- pinned for the scheduler;
    // scripts/check_rows_isa.py verifies it in the generated code of every variant)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RING; ++i) wreg[i] = wt[(size_t)min(i, kt_total - 1) * 64];
    __builtin_amdgcn_sched_barrier(0);
    // this wave's rows (and gamma / beta) have landed once at most the weight requests behind them are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RING) : "memory");

    // ---- LayerNorm of the wave's rows, in place in LDS (W/torch_model.py:25-27: fp32 two-pass statistics over the fp16 row,
    // eps 1e-5, affine, rounded to fp16 -- gemv_small.hip's arithmetic, statement for statement) --------------------------------
    if constexpr (LN) {
        // two rows per trip: their dependent chains (the element-order sums, the reductions) interleave; a row is converted to
        // fp32 once and stays in registers for the three passes; gamma / beta are read and converted once per pair of rows
        // (8-wave workgroups take one row per trip: a wave has only 4 rows there, and the kernel has to stay within 96 registers to
        // fit twice per SIMD beside the K/V streams' waves)
        constexpr int RPT = NW == 8 ? 1 : 2;
        static_assert(RB % RPT == 0, "rows per wave and rows per trip");
        const float fK = (float)p.K;
#pragma unroll 1
        for (int jr = 0; jr < RB; jr += RPT) {
            unsigned char* xr[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) xr[q] = s_x + (wid + (jr + q) * NW) * row_bytes;
            float xf[RPT][XP][8];
#pragma unroll
            for (int q = 0; q < RPT; ++q)
#pragma unroll
                for (int u = 0; u < XP; ++u) {
                    const half8v x = *(const half8v*)(xr[q] + (min(lane + 64 * u, 64 * np - 1) << 4));
#pragma unroll
                    for (int e = 0; e < 8; ++e) xf[q][u][e] = (float)x[e];
                }
            float mean[RPT], rstd[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                float sum = 0.f;
#pragma unroll
                for (int u = 0; u < XP; ++u) {
                    if (lane + 64 * u < pieces_per_row) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum += xf[q][u][e];
                    }
                }
                mean[q] = wave_sum_pre_mfma(sum) / fK;
            }
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                float sq = 0.f;
#pragma unroll
                for (int u = 0; u < XP; ++u) {
                    if (lane + 64 * u < pieces_per_row) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = xf[q][u][e] - mean[q]; sq += d * d; }
                    }
                }
                rstd[q] = rsqrtf(wave_sum_pre_mfma(sq) / fK + 1e-5f);
            }
#pragma unroll
            for (int u = 0; u < XP; ++u) {
                const int cu = min(lane + 64 * u, 64 * np - 1) << 4;
                const half8v gm = *(const half8v*)(s_g + cu), bt = *(const half8v*)(s_g + np * 1024 + cu);
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    half8v y;
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = (h16)((xf[q][u][e] - mean[q]) * rstd[q] * (float)gm[e] + (float)bt[e]);
                    if (lane + 64 * u < pieces_per_row) *(half8v*)(xr[q] + ((lane + 64 * u) << 4)) = y;
                }
            }
        }
    }
    __syncthreads();
    if (!wave_active) return;

    // mode 2 adds into the residual stream: this lane's 4 x MT values of x are requested HERE, before the K loop, so that the
    // epilogue is stores only (next to the other groups' streams a dependent read-modify-write round trip at the end of every
    // out / cout projection cost more than its K loop).  The loads are unconditional -- a run-time branch around a load makes
    // hipcc wait for each one -- other modes read a few bytes of the input block instead and drop them.
    const bool add_x = p.mode == 2;
    const h16* xsrc = add_x ? p.x + (nb * 16 + rl) : p.A + (lane & 7);
    const int xld = add_x ? p.ldx : 0;
    h16 xres[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) xres[mt][r] = xsrc[(size_t)min(row0 + mt * 16 + g * 4 + r, p.M - 1) * xld];
    // (the channel's scale and bias with them: read behind the K loop they were one more round trip at the end of every launch)
    // (unconditional loads, converted behind the K loop: a load behind a run-time test is waited for inside its branch)
    const int col = nb * 16 + rl;
    const bool has_scale = WB != 16 && p.scale != nullptr, has_bias = (p.mode == 1 || p.mode == 2) && p.bias != nullptr;
    const h16 sc_raw = *((has_scale ? p.scale : (const h16*)p.Wt) + (has_scale ? col : 0));
    const h16 bias_raw = *((has_bias ? p.bias : (const h16*)p.Wt) + (has_bias ? col : 0));

    // ---- K loop: RING tiles per round, every index a compile-time constant (register arrays indexed at run time live
    // in scratch memory); a tile's slot is refilled as soon as its MFMAs are issued ----------------------------------------
    float4v acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = float4v{0.f, 0.f, 0.f, 0.f};
    const unsigned char* xrow = s_x + rl * row_bytes;         // row rl of row tile 0; tile mt: + mt * 16 rows
    // one round = RING tiles; REFILL: request tile t + RING into the slot tile t leaves (every round but the last)
    auto round = [&](int t0, auto refill_tag) {
        constexpr bool REFILL = decltype(refill_tag)::value;
#pragma unroll
        for (int i = 0; i < RING; ++i) {
            const int t = t0 + i;
            const u32x4 wv4 = wreg[i];
            if constexpr (REFILL) wreg[i] = wt[(size_t)min(t + RING, kt_total - 1) * 64];
            if (t < kt_total) {                               // wave-uniform
                half8v b[NM];
                if constexpr (WB == 16) {
                    b[0] = __builtin_bit_cast(half8v, wv4);
                } else if constexpr (WB == 8) {
                    half2v h[8];
                    cvt_s8x4_f16x4(wv4.x, h[0], h[1]);
                    cvt_s8x4_f16x4(wv4.y, h[2], h[3]);
                    cvt_s8x4_f16x4(wv4.z, h[4], h[5]);
                    cvt_s8x4_f16x4(wv4.w, h[6], h[7]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        b[0][2 * q] = h[q][0]; b[0][2 * q + 1] = h[q][1];
                        b[1][2 * q] = h[4 + q][0]; b[1][2 * q + 1] = h[4 + q][1];
                    }
                } else {
                    const uint32_t wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
                    const half2v bias8 = {(h16)1032.0f, (h16)1032.0f};
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int sft = 0; sft < 4; ++sft) {
                            const uint32_t bits = ((wv[m] >> (4 * sft)) & 0x000F000Fu) | 0x64006400u;   // fp16 (1024 + u) x 2
                            const half2v pr = __builtin_bit_cast(half2v, bits) - bias8;                  // u - 8 = q, exact
                            b[m][2 * sft] = pr[0]; b[m][2 * sft + 1] = pr[1];
                        }
                }
                // this lane's KT / 4 inputs of the tile start at chunk t * (KT / 8) + g * NM
                const int c0 = t * (KT / 8) + g * NM;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int c = c0 + m;
                    const int off = c << 4;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const half8v a = *(const half8v*)(xrow + mt * 16 * row_bytes + off);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[m], acc[mt], 0, 0, 0);
                    }
                }
            }
        }
    };
    int t0 = 0;
    for (; t0 + RING < kt_total; t0 += RING) round(t0, std::true_type{});
    round(t0, std::false_type{});

    // ---- epilogue (epilogue.h: the rounding points of the row kernel) -------------------------------------------------------
    const float sc = has_scale ? (float)sc_raw : 1.0f;
    FusedEpilogue ep{p.mode, p.bias, p.gelu_kind, p.out32, p.ld32, p.out16, p.ld16, p.n_valid, p.x, p.ldx};
    const float bias2 = add_x && has_bias ? (float)bias_raw : 0.f;
    FusedEpiloguePre epre{};
    epre.bias_raw = bias_raw; epre.has_bias = has_bias;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = acc[mt][r] * sc;
        if (add_x) {                               // epilogue.h's mode 2 on the prefetched residual values: x = fp16(x + fp16(y + bias))
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + mt * 16 + g * 4 + r;
                if (row < p.M) p.x[(size_t)row * p.ldx + col] = (h16)r16((float)xres[mt][r] + r16(y[r] + bias2));
            }
        } else {
            fused_epilogue_tile(ep, p.M, nb, ms * MT + mt, lane, y, &epre);       // (modes 0 and 1 here: the residual values of `epre` are not read)
        }
    }
}

bool gemm_rows_supports(int K, int w8) {
    const int KT = w8 == 4 ? 128 : (w8 ? 64 : 32);
    return K % KT == 0 && K % 8 == 0 && K >= 8 && K <= 64 * 8 * rows::XP;
}

int launch_gemm_rows(const GemvSmallParams& p, hipStream_t stream) {
    using namespace rows;
    WM_REQUIRE(p.M >= 1, "gemm_rows: empty M");
    WM_REQUIRE(p.w8 == 0 || p.w8 == 1 || p.w8 == 4, "gemm_rows: w8=%d (0 fp16, 1 int8, 4 packed int4)", p.w8);
    WM_REQUIRE(gemm_rows_supports(p.K, p.w8), "gemm_rows: K=%d not supported (a multiple of the weight tile depth, <= %d)", p.K, 64 * 8 * XP);
    WM_REQUIRE(p.lda % 8 == 0, "gemm_rows: lda=%d must be a multiple of 8", p.lda);
    WM_REQUIRE(p.mode >= 0 && p.mode <= 2, "gemm_rows: mode=%d (0 fp32 sums, 1 bias + GELU, 2 residual add in place)", p.mode);
    WM_REQUIRE(p.mode != 0 || p.out32, "gemm_rows: mode 0 needs out32");
    WM_REQUIRE(p.mode != 1 || p.out16, "gemm_rows: mode 1 needs out16");
    WM_REQUIRE(p.mode != 2 || p.x, "gemm_rows: mode 2 needs x");
    WM_REQUIRE(!p.ln_g || p.ln_b, "gemm_rows: LayerNorm needs beta");
    const int n_full = p.n_blocks * 16;
    WM_REQUIRE(p.n_valid == 0 || p.n_valid == n_full, "gemm_rows: whole 16-column blocks only: n_valid=%d, 16 * n_blocks = %d", p.n_valid, n_full);
    WM_REQUIRE(p.mode != 0 || p.ld32 >= n_full, "gemm_rows: ld32=%d < 16 * n_blocks = %d", p.ld32, n_full);
    WM_REQUIRE(p.mode != 1 || p.ld16 >= n_full, "gemm_rows: ld16=%d < 16 * n_blocks = %d", p.ld16, n_full);
    WM_REQUIRE(p.mode != 2 || p.ldx >= n_full, "gemm_rows: ldx=%d < 16 * n_blocks = %d", p.ldx, n_full);
    // rows per workgroup: 32 (two MFMA row tiles) from 17 rows on; 8 waves (128 channels) per workgroup from 2560 channels on:
    // choices by the shapes only, and none of them touches a row's arithmetic
    static const int lab_mt = lab_env_int("WM_ROWS_MT", 0);      // A/B runs (WM_LAB=1): 1 | 2
    static const int lab_nw = lab_env_int("WM_ROWS_NW", 0);      // A/B runs (WM_LAB=1): 4 | 8
    const int MT = lab_mt == 1 ? 1 : (p.M > 16 ? 2 : 1);
    const int NW = lab_nw == 4 || lab_nw == 8 ? lab_nw : (p.n_blocks >= 160 ? 8 : 4);
    const int n_ms = (p.M + 16 * MT - 1) / (16 * MT), n_cg = (p.n_blocks + NW - 1) / NW;
    const int np = ((p.K >> 3) + 63) >> 6;
    const size_t lds = (size_t)16 * MT * (np * 1024 + 16) + (p.ln_g ? 2 * np * 1024 : 0);
    using Kern = void (*)(GemvSmallParams, int, int);
#define WM_ROWS_KERNS(WB) {{{gemm_rows_kernel<WB, 1, false, 4>, gemm_rows_kernel<WB, 1, false, 8>}, {gemm_rows_kernel<WB, 1, true, 4>, gemm_rows_kernel<WB, 1, true, 8>}}, \
                           {{gemm_rows_kernel<WB, 2, false, 4>, gemm_rows_kernel<WB, 2, false, 8>}, {gemm_rows_kernel<WB, 2, true, 4>, gemm_rows_kernel<WB, 2, true, 8>}}}
    static const Kern kerns[3][2][2][2] = {WM_ROWS_KERNS(16), WM_ROWS_KERNS(8), WM_ROWS_KERNS(4)};      // [weights][MT - 1][LN][NW == 8]
#undef WM_ROWS_KERNS
    // the dynamic-LDS limit is an attribute of the function ON A DEVICE (one process may drive several GPUs)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    WM_CHECK_HIP(hipGetDevice(&dev));
    const int slot = (dev >= 0 && dev < 64) ? dev : 0;
    if (slot != dev || !attr_set[slot].load(std::memory_order_acquire)) {
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c)
                    for (int d = 0; d < 2; ++d)
                        WM_CHECK_HIP(hipFuncSetAttribute((const void*)kerns[a][b][c][d], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[slot].store(true, std::memory_order_release);
    }
    const int grid = 8 * ((n_cg + 7) / 8) * n_ms;
    const Kern k = kerns[p.w8 == 4 ? 2 : (p.w8 ? 1 : 0)][MT - 1][p.ln_g ? 1 : 0][NW == 8 ? 1 : 0];
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW), lds, stream, p, n_cg, n_ms);
    WM_LAUNCH_CHECK(stream, "gemm_rows");
    return 0;
}

}  // namespace wm
