// Weight-streaming GEMM for SMALL decode batches (M = B * L <= 32 activation rows): the reference's own operating point
// is batch 1 (W/run.py:43-46, W/decoding.py:785-821), where a decode step is a chain of ~390 dependent launches and
// nothing but launch and memory latency.  This kernel removes a third of the chain: it does, in ONE launch, what the
// big-batch path spreads over the weight-streaming GEMM + the row kernel (+ the LayerNorm inside the row kernel):
//
//   [LayerNorm of the input rows, applied on the fly]  ->  x . W^T  (W read once, tile-linear, int8 / int4 / fp16)
//   -> K slices combined inside the workgroup  ->  bias / GELU / residual add / row statistics for the next LayerNorm
//
// Replaces, like gemm_skinny.hip: weight_only_gemv_launcher (weightOnlyMatrixVectorMultiplication.cu:136-277,371-378),
// the small-M branch of WeightOnlyQuantMatmulPlugin::enqueue (weightOnlyQuantMatmulPlugin.cpp:182-197), the fp16 MatMul of
// the non-quantised engines, the tied logits projection (whisper/model.py:290) -- and the element-wise layers around them
// (LayerNorm normalization.py:6-30, bias add quantization/layer.py:311-312, gelu functional.py:2044-2056, residual adds
// whisper/model.py:61-122).
//
// Structure.  One workgroup per 16 output channels; its waves (up to 16, at most 5 weight tiles = 5 KiB each) split K
// and meet in LDS -- no fp32 slabs in global memory, no second kernel.  A wave requests everything it needs from global
// memory at once, before it waits for anything: its weight tiles (HBM -> VGPRs in MFMA B-operand order, one wave-wide
// 16-byte load = 1 KiB contiguous) and its activation fragments (L2: a few KB per row shared by every workgroup, 16 rows
// x 32-64 bytes per load in MFMA A-operand order) -- the launch is ONE memory round trip plus the LDS meeting.
// The K slices are added in wave order; the number of waves depends on the weight shape only, so a row's sums do not
// depend on the batch it is in.  (They are not the big-batch path's sums bit for bit: that path cuts K into 4 slabs.)
// LayerNorm without a LayerNorm kernel: M <= 32 rows of 1280 channels are a few KB, so every workgroup copies the input
// rows to LDS, computes the two-pass fp32 statistics and normalises them in place itself ((x - mean) * rstd * gamma +
// beta, rounded to fp16: W/torch_model.py:25-27) while its weights are in flight; the MFMA A fragments are then read from
// LDS.  (A first version built the statistics from per-block partial sums left by the producing kernel: 20 dependent
// L2 round trips per row at the head of every projection -- B = 1 was SLOWER than the 12-launch chain, 2.50 vs 2.26 ms per
// token.)
#include <atomic>

#include "common.h"
#include "epilogue.h"
#include "kernels.h"

namespace wm {

// weight tiles (1 KiB) per wave -- a wave's whole share is requested at once: 5 int8 / int4 tiles (320 / 640 inputs), 10 fp16
// tiles (320 inputs); K = 5120 then takes the 16 waves a workgroup can have
constexpr int gemv_tpw(int wb) { return wb == 16 ? 10 : 5; }
constexpr int GEMV_MAX_WAVES = 16;

template <int WB, int MT, bool LN>       // WB: weight bits (16, 8, 4); MT: 16-row MFMA tiles (1, 2); LN: LayerNorm of the input rows
__global__ __launch_bounds__(LN ? 512 : 1024) void gemv_small_kernel(GemvSmallParams p) {
    constexpr int KT = WB == 4 ? 128 : (WB == 8 ? 64 : 32);   // inputs per 1 KiB weight tile
    constexpr int NM = KT / 32;                               // MFMAs (32-deep) per tile
    constexpr int TB = gemv_tpw(WB);
    __shared__ __attribute__((aligned(16))) float s_red[GEMV_MAX_WAVES][MT][64][4];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_x[];   // LayerNorm'ed input rows: [M][K + 8] fp16 (only with ln_g)

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, rl = lane & 15;
    const int nb = blockIdx.x, nthr = blockDim.x, nwave = nthr >> 6;
    const int kt_total = p.K / KT;
    const int tps = (kt_total + p.ksplit - 1) / p.ksplit;     // tiles per K slice (<= TB)
    const int t_begin = wid * tps, t_end = min(kt_total, t_begin + tps);
    const u32x4* wt = (const u32x4*)p.Wt + (size_t)nb * kt_total * 64 + lane;
    constexpr bool ln = LN;                                   // compile-time: run-time branches around the register arrays below demote them to scratch

    // ---- Everything this wave needs from global memory is requested up front, in the order it will be waited for (the
    // memory counter is in order: data requested behind the weights could only be waited for together with them):
    //   1. with LayerNorm: the rows whose statistics this wave computes (L2 hits, back first; the weights stay in flight
    //      behind them), and the gamma / beta fragments;   2. the wave's weight tiles (HBM, read once);
    //   3. the wave's activation fragments (L2) -- with LayerNorm requested after the statistics (registers).
    // (No load below sits behind a per-element run-time test: hipcc branches around such a load and waits for each one
    // before the next -- a dependent round trip per element.  Out-of-range elements re-read the last valid one instead.)
    // (the epilogue's per-channel scale, bias and residual values too: they used to be read behind the barrier that joins the
    // K slices, a round trip at the end of every launch)
    // -- all of them unconditional loads, converted where they are used (a load behind a run-time test is waited for inside its branch)
    const int col = nb * 16 + rl;
    const bool has_scale = WB != 16 && p.scale != nullptr;
    const h16 sc_raw = *((has_scale ? p.scale : (const h16*)p.Wt) + (has_scale ? col : 0));
    FusedEpilogue ep{p.mode, p.bias, p.gelu_kind, p.out32, p.ld32, p.out16, p.ld16, p.n_valid, p.x, p.ldx};
    FusedEpiloguePre epre[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) epre[mt] = fused_epilogue_prefetch(ep, p.M, nb, mt, lane, p.Wt);
    const int pieces_per_row = p.K >> 3;
    constexpr int RB = 8, XP = 3;               // rows per wave and sweep (register budget); 16-byte pieces per lane and row (K <= 1536)
    uint4 xv[LN ? RB : 1][LN ? XP : 1];
    auto stat_rows_load = [&](int r_first) {                  // rows r_first, r_first + nwave, ... (clamped)
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const h16* row = p.A + (size_t)min(r_first + j * nwave, p.M - 1) * p.lda;
#pragma unroll
            for (int u = 0; u < XP; ++u) xv[j][u] = *(const uint4*)(row + min(lane + 64 * u, pieces_per_row - 1) * 8);
        }
    };
    half8v gp[LN ? XP : 1], bp[LN ? XP : 1];                  // gamma / beta at this lane's pieces
    if constexpr (ln) {
        stat_rows_load(wid);
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            gp[u] = *(const half8v*)(p.ln_g + min(lane + 64 * u, pieces_per_row - 1) * 8);
            bp[u] = *(const half8v*)(p.ln_b + min(lane + 64 * u, pieces_per_row - 1) * 8);
        }
    }
    const int t_last = max(t_end - 1, 0);
    u32x4 wreg[TB];
#pragma unroll
    for (int i = 0; i < TB; ++i) wreg[i] = __builtin_nontemporal_load(wt + (size_t)min(t_begin + i, t_last) * 64);
    half8v a[TB][MT][NM];       // this lane's KT / 4 inputs of each tile, per row tile: NM fragments of 8 (rows >= M: clamped, never stored)
    auto frag_load = [&]() {
#pragma unroll
        for (int i = 0; i < TB; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h16* arow = p.A + (size_t)min(mt * 16 + rl, p.M - 1) * p.lda + (KT / 4) * g + (size_t)min(t_begin + i, t_last) * KT;
#pragma unroll
                for (int m = 0; m < NM; ++m) a[i][mt][m] = *(const half8v*)(arow + m * 8);
            }
    };
    if constexpr (!ln) frag_load();

    // ---- LayerNorm of the input rows (W/torch_model.py:25-27: fp32 statistics over the fp16 row, two passes, eps 1e-5,
    // affine, result rounded to fp16) without a LayerNorm kernel.  Row r belongs to wave r % nwave: the wave has the row in
    // registers (a lane holds pieces lane, lane + 64, lane + 128), runs both passes and the affine step there and writes
    // the normalised row to LDS once; after one barrier the MFMA fragments are read from LDS.  Everything from global memory
    // (rows, gamma / beta, weights) was requested before the first wait: one round trip.
    // (Two earlier forms, both slower: all rows copied to LDS raw and normalised there in three LDS passes -- the copy
    // took several dependent sweeps above 8 rows, 14.8 us per launch at 16 rows against 7.3 us without LayerNorm; and
    // statistics only, every lane normalising its own fragments after a second round trip to L2 -- +5 us at ONE row.)
    if constexpr (ln) {
        const int xs_row = (p.K + 8) * 2;                     // bytes; K + 8 halves: the 16 rows of a fragment read land on different banks
        for (int r0 = wid; r0 < p.M; r0 += RB * nwave) {
            if (r0 != wid) stat_rows_load(r0);
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                if (r0 + j * nwave >= p.M) continue;          // wave-uniform (the loads of such a row were clamped, its values are not used)
                float sum = 0.f;
#pragma unroll
                for (int u = 0; u < XP; ++u) {
                    const half8v x = __builtin_bit_cast(half8v, xv[j][u]);
                    if (lane + 64 * u < pieces_per_row) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum += (float)x[e];
                    }
                }
                const float mean = wave_sum_pre_mfma(sum) / (float)p.K;
                float sq = 0.f;
#pragma unroll
                for (int u = 0; u < XP; ++u) {
                    const half8v x = __builtin_bit_cast(half8v, xv[j][u]);
                    if (lane + 64 * u < pieces_per_row) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = (float)x[e] - mean; sq += d * d; }
                    }
                }
                const float rstd = rsqrtf(wave_sum_pre_mfma(sq) / (float)p.K + 1e-5f);
                const int r = r0 + j * nwave;
#pragma unroll
                for (int u = 0; u < XP; ++u) {
                    half8v x = __builtin_bit_cast(half8v, xv[j][u]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = (h16)(((float)x[e] - mean) * rstd * (float)gp[u][e] + (float)bp[u][e]);
                    if (lane + 64 * u < pieces_per_row) *(half8v*)(s_x + r * xs_row + (lane + 64 * u) * 16) = x;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TB; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h16* arow = (const h16*)(s_x + min(mt * 16 + rl, p.M - 1) * xs_row) + (KT / 4) * g + (size_t)min(t_begin + i, t_last) * KT;
#pragma unroll
                for (int m = 0; m < NM; ++m) a[i][mt][m] = *(const half8v*)(arow + m * 8);
            }
    }

    float4v acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = float4v{0.f, 0.f, 0.f, 0.f};
    // (fully unrolled with compile-time indices: a `break` here once kept the loop rolled, which put wreg[] and a[] -- indexed
    // by a run-time i -- into scratch memory: every GEMV took 6 us instead of 3 and the logits 105 us instead of 30)
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        const bool valid = t_begin + i < t_end;      // wave-uniform
        // the tile's weights as NM fp16 B operands (exact expansions, as gemm_skinny.hip)
        half8v b[NM];
        if constexpr (WB == 16) {
            b[0] = __builtin_bit_cast(half8v, wreg[i]);
        } else if constexpr (WB == 8) {
            half2v h[8];
            cvt_s8x4_f16x4(wreg[i].x, h[0], h[1]);
            cvt_s8x4_f16x4(wreg[i].y, h[2], h[3]);
            cvt_s8x4_f16x4(wreg[i].z, h[4], h[5]);
            cvt_s8x4_f16x4(wreg[i].w, h[6], h[7]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b[0][2 * j] = h[j][0]; b[0][2 * j + 1] = h[j][1];
                b[1][2 * j] = h[4 + j][0]; b[1][2 * j + 1] = h[4 + j][1];
            }
        } else {
            const uint32_t wv[4] = {wreg[i].x, wreg[i].y, wreg[i].z, wreg[i].w};
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
        if (valid) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][mt][m], b[m], acc[mt], 0, 0, 0);
        }
    }

    // ---- K slices meet in LDS (each already multiplied by the per-channel scale, as the big-batch path's slabs are) and
    // are added in wave order by wave 0, which also runs the epilogue -----------------------------------------------------
    const float sc = has_scale ? (float)sc_raw : 1.0f;
    if (nwave > 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float4v v = acc[mt];
            v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc;
            *(float4v*)&s_red[wid][mt][lane][0] = v;
        }
        __syncthreads();
        if (wid != 0) return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float y[4];
        if (nwave > 1) {
            // (the waves' sums are read FOUR at a time -- in flight together -- and added in wave order: read and added one by one they are up
            // to sixteen dependent LDS round trips; four at a time keeps the kernel inside its 128 registers.  csrc/gemv_chain.hip's
            // epilogue reads all of them at once, profiles/r5v_*)
            float4v s;
            if constexpr (WB == 4 || MT > 1) {                    // (the int4 and the two-row-tile forms have no registers to spare: one at a time)
                s = *(const float4v*)&s_red[0][mt][lane][0];
                for (int w = 1; w < nwave; ++w) {
                    const float4v tw = *(const float4v*)&s_red[w][mt][lane][0];
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] += tw[r];
                }
            } else {
                s = float4v{0.f, 0.f, 0.f, 0.f};
                for (int w0 = 0; w0 < nwave; w0 += 4) {
                    float4v part[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) part[k] = *(const float4v*)&s_red[w0 + k < nwave ? w0 + k : 0][mt][lane][0];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (w0 + k < nwave) {
                            if (w0 + k == 0) s = part[0];
                            else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) s[r] += part[k][r];
                            }
                        }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = s[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = acc[mt][r] * sc;
        }
        fused_epilogue_tile(ep, p.M, nb, mt, lane, y, &epre[mt]);
    }
}

int gemv_small_waves(int K, int w8) {                // K slices = waves per workgroup: at most GEMV_TPW weight tiles per wave
    const int KT = w8 == 4 ? 128 : (w8 ? 64 : 32);
    const int kt_total = K / KT;
    const int tpw = gemv_tpw(w8 == 4 ? 4 : (w8 ? 8 : 16));
    int n = (kt_total + tpw - 1) / tpw;
    return n < 1 ? 1 : n;
}

template <int WB>
static int launch_mt(const GemvSmallParams& p, hipStream_t stream) {
    const dim3 grid(p.n_blocks), block(64 * p.ksplit);
    const size_t lds = p.ln_g ? (size_t)p.M * (p.K + 8) * 2 : 0;          // the LayerNorm'ed input rows
    if (p.ln_g) {
        if (p.M <= 16) hipLaunchKernelGGL((gemv_small_kernel<WB, 1, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((gemv_small_kernel<WB, 2, true>), grid, block, lds, stream, p);
    } else {
        if (p.M <= 16) hipLaunchKernelGGL((gemv_small_kernel<WB, 1, false>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((gemv_small_kernel<WB, 2, false>), grid, block, 0, stream, p);
    }
    return 0;
}

int launch_gemv_small(const GemvSmallParams& p_, hipStream_t stream) {
    GemvSmallParams p = p_;
    WM_REQUIRE(p.M >= 1 && p.M <= GEMV_SMALL_MAX_M, "gemv_small: M=%d out of range [1,%d]", p.M, GEMV_SMALL_MAX_M);
    WM_REQUIRE(p.w8 == 0 || p.w8 == 1 || p.w8 == 4, "gemv_small: w8=%d (0 fp16, 1 int8, 4 packed int4)", p.w8);
    const int KT = p.w8 == 4 ? 128 : (p.w8 ? 64 : 32);
    WM_REQUIRE(p.K % KT == 0, "gemv_small: K=%d must be a multiple of %d", p.K, KT);
    WM_REQUIRE(p.lda % 8 == 0, "gemv_small: lda=%d must be a multiple of 8", p.lda);
    p.ksplit = gemv_small_waves(p.K, p.w8);          // a property of the weight shape only: a row's sums do not depend on the batch
    WM_REQUIRE(p.ksplit <= GEMV_MAX_WAVES, "gemv_small: K=%d needs %d waves per workgroup (max %d)", p.K, p.ksplit, GEMV_MAX_WAVES);
    WM_REQUIRE(p.mode >= 0 && p.mode <= 3, "gemv_small: mode=%d", p.mode);
    WM_REQUIRE(p.mode != 0 || p.out32, "gemv_small: mode 0 needs out32");
    WM_REQUIRE((p.mode != 1 && p.mode != 3) || p.out16, "gemv_small: modes 1 and 3 need out16");
    WM_REQUIRE(p.mode != 2 || p.x, "gemv_small: mode 2 needs x");
    // LayerNorm variant: a lane holds up to 3 16-byte pieces of a row, a workgroup has at most 8 waves (register budget),
    // the normalised rows sit in LDS
    constexpr size_t LN_LDS_MAX = 100 * 1024;      // 32 rows of 1536 channels
    WM_REQUIRE(!p.ln_g || (p.ln_b && p.K % 8 == 0 && p.K <= 1536 && p.ksplit <= 8 && (size_t)p.M * (p.K + 8) * 2 <= LN_LDS_MAX),
               "gemv_small: LayerNorm needs beta, K <= 1536 and <= 8 K slices (K=%d, slices=%d)", p.K, p.ksplit);
    // modes 0-2 read bias[col] and write out32 / out16 / x for every column of every 16-wide block (only the logits mode
    // has a ragged last block): the logical width must be the blocks' width
    const int n_full = p.n_blocks * 16;
    WM_REQUIRE(p.mode == 3 || p.n_valid == 0 || p.n_valid == n_full,
               "gemv_small: mode %d writes whole 16-column blocks: n_valid=%d must be 16 * n_blocks = %d", p.mode, p.n_valid, n_full);
    WM_REQUIRE(p.mode != 0 || p.ld32 >= n_full, "gemv_small: ld32=%d < 16 * n_blocks = %d", p.ld32, n_full);
    WM_REQUIRE(p.mode != 1 || p.ld16 >= n_full, "gemv_small: ld16=%d < 16 * n_blocks = %d", p.ld16, n_full);
    WM_REQUIRE(p.mode != 2 || p.ldx >= n_full, "gemv_small: ldx=%d < 16 * n_blocks = %d", p.ldx, n_full);
    // the dynamic-LDS limit is an attribute of the function ON A DEVICE: one process may drive several GPUs from several threads
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    WM_CHECK_HIP(hipGetDevice(&dev));
    const int slot = (dev >= 0 && dev < 64) ? dev : 0;
    if (slot != dev || !attr_set[slot].load(std::memory_order_acquire)) {       // (devices past the table: set it every time)
        const void* kerns[6] = {(const void*)gemv_small_kernel<16, 1, true>, (const void*)gemv_small_kernel<16, 2, true>, (const void*)gemv_small_kernel<8, 1, true>,
                                (const void*)gemv_small_kernel<8, 2, true>, (const void*)gemv_small_kernel<4, 1, true>, (const void*)gemv_small_kernel<4, 2, true>};
        for (const void* k : kerns) WM_CHECK_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LN_LDS_MAX));
        attr_set[slot].store(true, std::memory_order_release);
    }
    if (p.w8 == 4) launch_mt<4>(p, stream); else if (p.w8) launch_mt<8>(p, stream); else launch_mt<16>(p, stream);
    WM_LAUNCH_CHECK(stream, "gemv_small");
    return 0;
}

}  // namespace wm
