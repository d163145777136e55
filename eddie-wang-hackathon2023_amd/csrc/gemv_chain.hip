// The decoder of a group of ONE TO EIGHT rows (utterances with one new token each; one row is the reference's own operating point,
// batch 1: W/run.py:43-46, W/decoding.py:785-821) as the stages of one launch: a decoder layer is
//
//     self-attention (cache append included)  ->  out projection + residual  ->  LayerNorm + cross-attention query projection  ->
//     cross-attention over 4 key-range pieces  ->  merge of the pieces + cross-attention out projection + residual  ->
//     LayerNorm + mlp1 + GELU  ->  mlp2 + residual  ->  LayerNorm + qkv of the next layer
//
// and the launch walks over the layers itself (or serves one layer per launch): 291 launches per token step become 4.  Replaces
// the same reference code as gemv_small.hip (weight_only_gemv_launcher, weightOnlyMatrixVectorMultiplication.cu:136-277; the
// small-M branch of WeightOnlyQuantMatmulPlugin::enqueue; the element-wise layers around the Linears, whisper/model.py:61-122) and
// as attn_decode.hip's kernels (MaskedMultiheadAttention, decoderMaskedMultiheadAttentionTemplate.h:1195-2188, for the self- and the
// cross-attention) -- with the arithmetic of gemv_small_kernel, attn_self_wg_kernel, attn_cross_kernel<1> and
// attn_cross_combine_kernel, bit for bit (tests/test_gpu_round4.py).
//
// Why, and why only one row.  At batch 1 a token step is ~ 300 dependent launches of ~ 5.5 us: ~ 2.5 us of kernel boundary and
// ~ 3 us of body, most of the body the round trip of the weights.  Rounds 2-3 removed launches by REDUNDANT recomputation and
// gained nothing; every hand-off between workgroups inside a launch they tried went through fences or flags
// and was slower than the boundary.  scripts/lab/edge_lab.hip (profiles/r4d_*, r4k_*) measured the one form that is not: the
// activation vector as 8-byte {epoch, value} GRANULES, each written by one write-through (sc1) store, every consumer sweeping
// the whole vector with 16-byte sc1 loads (all in flight, one wait) until every tag carries the stage's epoch -- no flag, no
// fence, no barrier: 1.9 us per all-to-all edge for 2 048 halves, 2.6 us for 5 120, against 2.0-4.2 us for a kernel boundary
// around a trivial body.  The edge grows with rows x width (4 rows: 3.5 us, 16 rows of 8-byte loads: 25 us) where a boundary
// does not, so the launch serves ONE row.  What it wins is everything around the edges: a stage's weights (and scale, bias,
// LayerNorm vectors) are requested BEFORE its input is waited for; the cross-attention's K / V rows and the self-attention's
// cached rows are in LDS (by DMA, a stage or a layer ahead) before q exists; nothing a stage must have before it can request
// anything else -- its descriptor, the per-layer pointers -- is read from memory (they are copied to LDS once per launch: each
// such load was a microsecond in front of EVERY stage).  DESIGN.md section 5 "Batch 1" has the road and the stage timings.
//
// Structure.  256 workgroups of eight waves, one per CU, alive for the whole launch.  A Linear stage with K <= 4 x 320 inputs is
// run by "slots" of four waves (a slot = one group of 16 output channels = gemv_small's workgroup: its waves split K as there,
// meet in LDS in wave order), two slots per workgroup; K = 4 n_state (mlp2) by all eight waves as one slot, two K slices per
// wave, the sixteen slices added in slice order as gemv_small's sixteen waves are.  The workgroup that owns channels 16 c ..
// 16 c + 15 of the residual stream owns them in every stage (its copy lives in LDS), so the residual adds need no exchange.
// The attention stages sit on workgroups that idle through most Linear stages: a self-attention head on each of the 20 before
// the last 80, a cross-attention (head, piece) on each of the last 80 -- their upper four waves carry the DMA requests.
// Every wait is bounded: a wave that gives up sets *err and the rest of the launch falls through (the host checks the word).
// Rows (template NR): 1 and 2 rows as above, both utterances' K / V pieces in LDS; NR = 4 (3-4 rows) and 8 (5-8 rows, 8- and 16-bit
// weights) keep the cross-attention's rows in registers, two (row, head, piece) items of four waves per workgroup (chain_cross_stage4),
// and carry rows 4-7 in the second lane group of the MFMA's output; rows that have finished (p.live) read and append nothing.
#include <atomic>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace wm {

typedef __attribute__((address_space(1))) unsigned long long chain_gu64;
// The stage descriptors are read from memory, so the compiler knows nothing about the pointers in them and would use FLAT loads
// (both wait counters, the slower path) for the weights: they are device memory, and said to be so.
#define CHAIN_GLOBAL(T, ptr) ((const __attribute__((address_space(1))) T*)(ptr))

constexpr int CHAIN_MAX_IN = 4 * 1536;     // widest stage input (halves): mlp2's K = 4 n_state

// (the error word lives in pinned host memory the device reaches directly: the host reads it without a copy or a synchronisation --
// engine.hip: ChainDev -- so it is read and written at system scope; it is looked at once per 64 polls only)
__device__ __forceinline__ bool chain_failed(const unsigned* err) {
    return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
}

// NL 16-byte sc1 loads per lane (two granules each) from granule indices first[k] (even), repeated until every tag is `epoch`.
// All loads of a pass are in flight together; the destinations are named in the wait statement (cdna_hip_programming.md 5.7,
// form (ii)).  Returns false when the wait was given up.
template <int NL>
__device__ __forceinline__ bool sweep_granules16(const unsigned long long* gran, const int (&first)[NL], unsigned epoch, u32x4 (&val)[NL],
                                                 unsigned* err, int lane) {
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int k = 0; k < NL; ++k)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(val[k]) : "v"(gran + first[k]) : "memory");
        if constexpr (NL == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]) :: "memory");
        else if constexpr (NL == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]) :: "memory");
        else if constexpr (NL == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]) :: "memory");
        else if constexpr (NL == 8)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]) :: "memory");
        else if constexpr (NL == 12)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]),
                         "+v"(val[8]), "+v"(val[9]), "+v"(val[10]), "+v"(val[11]) :: "memory");
        else if constexpr (NL == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]) :: "memory");
        else if constexpr (NL == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]) :: "memory");
        else if constexpr (NL == 18)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]),
                         "+v"(val[8]), "+v"(val[9]), "+v"(val[10]), "+v"(val[11]), "+v"(val[12]), "+v"(val[13]), "+v"(val[14]), "+v"(val[15]),
                         "+v"(val[16]), "+v"(val[17]) :: "memory");
        else if constexpr (NL == 16)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]),
                         "+v"(val[8]), "+v"(val[9]), "+v"(val[10]), "+v"(val[11]), "+v"(val[12]), "+v"(val[13]), "+v"(val[14]), "+v"(val[15]) :: "memory");
        else if constexpr (NL == 9)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]),
                         "+v"(val[8]) :: "memory");
        else if constexpr (NL == 15)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]),
                         "+v"(val[8]), "+v"(val[9]), "+v"(val[10]), "+v"(val[11]), "+v"(val[12]), "+v"(val[13]), "+v"(val[14]) :: "memory");
        else static_assert(NL == 1 || NL == 2 || NL == 3 || NL == 4 || NL == 6 || NL == 8 || NL == 9 || NL == 12 || NL == 15 || NL == 16 || NL == 18, "sweep sizes of the chain");
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NL; ++k) ok &= val[k].y == epoch && val[k].w == epoch;
        if (__all(ok)) return true;
        if ((spins & 63) == 63 && chain_failed(err)) return false;
        if (spins > (1u << 20)) {                     // ~ a second: a workgroup of the chain is not running
            if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// one stage of the chain.  WIDE: one slot of eight waves, two K slices per wave (K = 4 n_state); LN: LayerNorm of the
// residual-stream row in the prologue.
// Where a stage without LayerNorm takes its input row from (wave-uniform):
enum ChainIn : int {
    CHAIN_IN_LDS = 0,          // the row is in s_in[0] already (chain_merge_tagged has put it there)
    CHAIN_IN_GRANULES = 1      // an fp16 row published as granules in this launch: `gran`, tagged `tag`
};
template <int NR>
__device__ __forceinline__ void chain_merge_tagged(const GemvChainParams& p, unsigned tag, h16 (*s_in)[CHAIN_MAX_IN + 8]);      // (below)

// NR activation rows (1 | 2 | 4: the kernel for 3 and 4 rows; p.rows says how many exist).  The MFMA's 16 A rows carry the activation rows
// alternately (A row = lane & 15, row = lane & (NR - 1)), so a row's sums are the same chain of MFMA steps whatever NR is, and come out in
// accumulator element `row` of the first 16 lanes.  (With 3 rows the fourth A row multiplies whatever LDS holds: its sums are never read.)
// Where the rows of a stage's input sit in LDS (s_in is max(NR, 2) arrays of CHAIN_MAX_IN + 8 halves):
//   NR == 1: s_in[slot] -- every slot keeps its own copy (no barrier between a slot's sweep and its reads)
//   NR >= 2: rows the whole workgroup shares (LayerNorm output, merged attention rows, the wide stage's hidden rows): s_in[row];
//            rows a slot sweeps for itself (K <= 1536): 2 NR blocks of 1544 halves, block NR * slot + row
template <int NR, bool WIDE, bool SHARED>
__device__ __forceinline__ h16* chain_in_row(h16 (*s_in)[CHAIN_MAX_IN + 8], int slot, int row) {
    if constexpr (NR == 1) return &s_in[slot][0];
    else if constexpr (WIDE || SHARED) return &s_in[row][0];
    else return &s_in[0][0] + (NR * slot + row) * 1544;
}

// gemv_small's LayerNorm of ONE row by one wave: the row in registers (a lane: pieces lane, lane + 64, lane + 128 of 8 halves), two-pass
// fp32 statistics, affine step, the normalised row to `dst` in LDS -- same operations, same order.  The row comes from plain memory
// (`xrow`: as the launches before this one left it) or from the granules the previous stage's owners have just published (`gx`, `tag`).
__device__ __forceinline__ void chain_ln_row(const GemvChainParams& p, const ChainStage& st, const h16* xrow, const unsigned long long* gx, unsigned tag,
                                             bool x_in_granules, h16* dst, int lane) {
    constexpr int XP = 3;
    const int pieces_per_row = st.K >> 3;
    half8v xr[XP], gp[XP], bp[XP];
#pragma unroll
    for (int u = 0; u < XP; ++u) {
        gp[u] = *CHAIN_GLOBAL(half8v, st.ln_g + min(lane + 64 * u, pieces_per_row - 1) * 8);
        bp[u] = *CHAIN_GLOBAL(half8v, st.ln_b + min(lane + 64 * u, pieces_per_row - 1) * 8);
    }
    bool ok = true;
    if (!x_in_granules) {                             // the row as the launches before this one left it (plain memory)
#pragma unroll
        for (int u = 0; u < XP; ++u) xr[u] = *(const half8v*)(xrow + min(lane + 64 * u, pieces_per_row - 1) * 8);
    } else {                                          // the row the previous stage's owners have just published
        int first[2 * XP];
        u32x4 val[2 * XP];
#pragma unroll
        for (int u = 0; u < XP; ++u) {                // piece q = granules 4 q .. 4 q + 3 = two 16-byte loads
            const int q = min(lane + 64 * u, pieces_per_row - 1);
            first[2 * u] = 4 * q; first[2 * u + 1] = 4 * q + 2;
        }
        ok = sweep_granules16<2 * XP>(gx, first, tag, val, p.err, lane);
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const u32x4 v = u32x4{val[2 * u].x, val[2 * u].z, val[2 * u + 1].x, val[2 * u + 1].z};
            xr[u] = __builtin_bit_cast(half8v, v);
        }
    }
    if (ok) {
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < XP; ++u)
            if (lane + 64 * u < pieces_per_row) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += (float)xr[u][e];
            }
        const float mean = wave_sum_pre_mfma(sum) / (float)st.K;
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < XP; ++u)
            if (lane + 64 * u < pieces_per_row) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xr[u][e] - mean; sq += d * d; }
            }
        const float rstd = rsqrtf(wave_sum_pre_mfma(sq) / (float)st.K + 1e-5f);
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            half8v x = xr[u];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (h16)(((float)x[e] - mean) * rstd * (float)gp[u][e] + (float)bp[u][e]);
            if (lane + 64 * u < pieces_per_row) *(half8v*)(dst + (lane + 64 * u) * 8) = x;
        }
    }
}

template <int WB, bool WIDE, bool LN, int NR>
__device__ __forceinline__ void chain_stage(const GemvChainParams& p, const ChainStage& st, int s, unsigned epoch, bool& own_valid,
                                            float (*s_red)[64][4], h16 (*s_in)[CHAIN_MAX_IN + 8], h16 (*s_own)[16],
                                            int in_kind, const unsigned long long* gran, unsigned tag, bool x_in_granules) {
    // in_kind == CHAIN_IN_LDS: the stage's input rows are the merge of the cross-attention's pieces (tagged `tag`), which the workgroup
    // performs INSIDE this stage, behind the weight requests (round 5: merged first and requested afterwards, the weights' round trip
    // -- ~ 1 us per layer -- sat between the arrival of the last piece and the multiply)
    constexpr int KT = WB == 4 ? 128 : (WB == 8 ? 64 : 32);   // inputs per 1 KiB weight tile
    constexpr int NM = KT / 32;
    constexpr int TB = WB == 16 ? 10 : 5;
    constexpr int NS = WIDE ? 2 : 1;                          // K slices per wave
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, rl = lane & 15;
    const int kt_total = st.K / KT;
    const int slices = (kt_total + TB - 1) / TB;
    const int tps = (kt_total + slices - 1) / slices;         // tiles per K slice (gemv_small: ksplit = slices)
    const int slot = WIDE ? 0 : wid >> 2, wslot = WIDE ? wid : wid & 3;
    const int grp = blockIdx.x + slot * gridDim.x;            // this slot's group of 16 output channels
    const bool has_group = grp < st.n_blocks;
    const int nb = has_group ? grp : st.n_blocks - 1;         // (idle slots re-read valid memory; nothing of theirs is stored)
    const int n_out = st.n_blocks * 16;                       // output channels of the stage = elements between the rows of its outputs
    const int row_a = NR == 1 ? 0 : rl & (NR - 1);            // the activation row this lane's A fragments carry
    const int R = NR == 1 ? 1 : p.rows;                       // rows that exist (NR == 4: 3 or 4)

    // an idle slot (no group of this stage falls to it) only keeps the workgroup's barriers company: it touches no memory, so that
    // waves which carry LDS-DMA requests for a later stage (the cross-attention's K / V rows) are not made to wait for them here
    if (!has_group) {
        if (!LN && !WIDE && in_kind == CHAIN_IN_LDS && (int)blockIdx.x < st.n_blocks) chain_merge_tagged<NR>(p, tag, s_in);      // (all eight waves merge)
        if constexpr (LN && NR == 8) {               // rows 4-7 are normalised by waves 4-7 -- this slot's, when the workgroup's other slot has a group
            if ((int)blockIdx.x < st.n_blocks && wid < R)
                chain_ln_row(p, st, p.x + wid * st.K, p.gran_x + wid * (st.K >> 1), epoch - 1, x_in_granules, chain_in_row<NR, WIDE, true>(s_in, 0, wid), lane);
        }
        const int nbar = LN ? 3 : 2;
        for (int b = 0; b < nbar; ++b) __syncthreads();
        if (st.mode == 2) own_valid = true;
        return;
    }

    // ---- 1. every weight tile of this wave's K slices is requested now, before anything is waited for -- and the epilogue's
    // per-channel scale and bias with them (requested in the epilogue they were a memory round trip of their own) -------------
    // (UNCONDITIONAL loads, converted where they are used: behind a run-time test hipcc branches around the load and waits for it
    // inside the branch -- two round trips in a row at the head of the stage, ahead of the weight requests)
    const int col = nb * 16 + rl;
    const bool has_scale = WB != 16 && st.scale != nullptr, has_bias = (st.mode == 1 || st.mode == 2) && st.bias != nullptr;
    const h16 sc_raw = *(CHAIN_GLOBAL(h16, has_scale ? (const void*)st.scale : st.Wt) + (has_scale ? col : 0));
    const h16 bias_raw = *(CHAIN_GLOBAL(h16, has_bias ? (const void*)st.bias : st.Wt) + (has_bias ? col : 0));
    u32x4 wreg[NS][TB];
    int t_begin[NS], t_end[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int slice = wslot + 8 * j;
        t_begin[j] = min(slice, slices - 1) * tps;
        t_end[j] = (slice < slices && has_group) ? min(kt_total, t_begin[j] + tps) : t_begin[j];     // (an absent slice multiplies nothing)
        const __attribute__((address_space(1))) u32x4* wt = CHAIN_GLOBAL(u32x4, st.Wt) + (size_t)nb * kt_total * 64 + lane;
        const int t_last = max(min(kt_total, t_begin[j] + tps) - 1, 0);
        if (slice < slices && has_group) {                    // (wave-uniform: an idle slot or an absent slice streams nothing)
#pragma unroll
            for (int i = 0; i < TB; ++i) wreg[j][i] = __builtin_nontemporal_load(wt + (size_t)min(t_begin[j] + i, t_last) * 64);
        } else {
#pragma unroll
            for (int i = 0; i < TB; ++i) wreg[j][i] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    half8v a[NS][TB][NM];

    // ---- 2. the stage's input row ------------------------------------------------------------------------------------------
    if constexpr (LN) {
        // NR == 1: wave 0 of the slot normalises the row for its slot.  NR == 2: waves 0 and 1 of the WORKGROUP (slot 0's: they never
        // carry LDS-DMA requests) normalise rows 0 and 1, once, for both slots
        if constexpr (NR == 1) {
            if (wslot == 0) chain_ln_row(p, st, p.x, p.gran_x, epoch - 1, x_in_granules, chain_in_row<NR, WIDE, true>(s_in, slot, 0), lane);
        } else {
            if (wid < R) chain_ln_row(p, st, p.x + wid * st.K, p.gran_x + wid * (st.K >> 1), epoch - 1, x_in_granules, chain_in_row<NR, WIDE, true>(s_in, 0, wid), lane);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            const int t_last = max(t_end[0] - 1, t_begin[0]);
            const h16* arow = chain_in_row<NR, WIDE, true>(s_in, slot, row_a) + (KT / 4) * g + (size_t)min(t_begin[0] + i, t_last) * KT;
#pragma unroll
            for (int m = 0; m < NM; ++m) a[0][i][m] = *(const half8v*)(arow + m * 8);
        }
    } else if (!WIDE && in_kind == CHAIN_IN_LDS) {
        // the merged attention rows go to s_in[row] (chain_merge_tagged, all eight waves, ended by a barrier) -- with this stage's weights
        // already requested
        chain_merge_tagged<NR>(p, tag, s_in);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int t_last = max(t_end[j] - 1, t_begin[j]);
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                const h16* arow = &s_in[row_a][0] + (KT / 4) * g + (size_t)min(t_begin[j] + i, t_last) * KT;
#pragma unroll
                for (int m = 0; m < NM; ++m) a[j][i][m] = *(const half8v*)(arow + m * 8);
            }
        }
    } else {
        // the hidden rows the previous stage has just published (mode 1): every wave sweeps the granules of ITS K slices into LDS and reads
        // its fragments back -- no barrier, its own data.  ONE pass for all of the wave's slices and rows (round 5: the wide stage's two
        // slices were two passes, a fabric round trip each: its input wait was 2.5 us against 1.1-1.5 for the other stages), and only as
        // many 16-byte loads per slice and row as a slice can need (tps x KT halves: 2 for 8- and 16-bit weights, 3 for 4-bit; the third
        // load of every pass re-read one granule 64 times)
        constexpr int NLR = (TB * KT / 4 <= 128) ? 2 : 3;                     // loads per lane, slice and row
        const int in_stride = st.K >> 1;                                      // granules between the rows of the stage's input
        constexpr int RP = (NR > 4 && WIDE) ? 4 : NR, NPASS = NR / RP;        // rows per pass: 16 loads in flight at most (NR == 8: the wide stage takes two passes of four rows)
        int n_ld[NS];
        bool any = false;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            n_ld[j] = (t_end[j] - t_begin[j]) * KT / 4;                       // 16-byte loads (4 halves each) of this slice (0: absent)
            any |= n_ld[j] > 0;
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            if (ps * RP >= R) break;                                          // (uniform) rows that do not exist
            int first[NS * RP * NLR];
            u32x4 val[NS * RP * NLR];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int g0 = t_begin[j] * KT / 2;                           // first granule
#pragma unroll
                for (int r = 0; r < RP; ++r)
#pragma unroll
                    for (int k = 0; k < NLR; ++k)
                        first[(j * RP + r) * NLR + k] = min(ps * RP + r, R - 1) * in_stride + g0 + 2 * min(lane + 64 * k, max(n_ld[j] - 1, 0));      // (a row that does not exist: the last one again)
            }
            const bool ok = any && sweep_granules16<NS * RP * NLR>(gran, first, tag, val, p.err, lane);      // (an absent slice re-reads granules of the stage's first tile)
            if (ok) {
#pragma unroll
                for (int j = 0; j < NS; ++j)
#pragma unroll
                    for (int r = 0; r < RP; ++r)
#pragma unroll
                        for (int k = 0; k < NLR; ++k)
                            if (lane + 64 * k < n_ld[j])
                                *(uint2*)(chain_in_row<NR, WIDE, false>(s_in, slot, ps * RP + r) + t_begin[j] * KT + (lane + 64 * k) * 4) =
                                    make_uint2(val[(j * RP + r) * NLR + k].x, val[(j * RP + r) * NLR + k].z);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // this wave's LDS writes before its reads
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int t_last = max(t_end[j] - 1, t_begin[j]);
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                const h16* arow = chain_in_row<NR, WIDE, false>(s_in, slot, row_a) + (KT / 4) * g + (size_t)min(t_begin[j] + i, t_last) * KT;
#pragma unroll
                for (int m = 0; m < NM; ++m) a[j][i][m] = *(const half8v*)(arow + m * 8);
            }
        }
    }

    // ---- 3. multiply (gemv_small's loop), K slices to LDS scaled, in slice order ------------------------------------------
    const float sc = has_scale ? (float)sc_raw : 1.0f;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        float4v acc = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            const bool valid = t_begin[j] + i < t_end[j];     // wave-uniform
            half8v b[NM];
            if constexpr (WB == 16) {
                b[0] = __builtin_bit_cast(half8v, wreg[j][i]);
            } else if constexpr (WB == 8) {
                half2v h[8];
                cvt_s8x4_f16x4(wreg[j][i].x, h[0], h[1]);
                cvt_s8x4_f16x4(wreg[j][i].y, h[2], h[3]);
                cvt_s8x4_f16x4(wreg[j][i].z, h[4], h[5]);
                cvt_s8x4_f16x4(wreg[j][i].w, h[6], h[7]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    b[0][2 * q] = h[q][0]; b[0][2 * q + 1] = h[q][1];
                    b[1][2 * q] = h[4 + q][0]; b[1][2 * q + 1] = h[4 + q][1];
                }
            } else {
                const uint32_t wv[4] = {wreg[j][i].x, wreg[j][i].y, wreg[j][i].z, wreg[j][i].w};
                const half2v bias8 = {(h16)1032.0f, (h16)1032.0f};
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int sft = 0; sft < 4; ++sft) {
                        const uint32_t bits = ((wv[m] >> (4 * sft)) & 0x000F000Fu) | 0x64006400u;
                        const half2v pr = __builtin_bit_cast(half2v, bits) - bias8;
                        b[m][2 * sft] = pr[0]; b[m][2 * sft + 1] = pr[1];
                    }
            }
            if (valid) {
#pragma unroll
                for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j][i][m], b[m], acc, 0, 0, 0);
            }
        }
        const int slice = wslot + 8 * j;
        if (slice < slices) {
            float4v v = acc;
            v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc;
            *(float4v*)&s_red[(WIDE ? 0 : 4 * slot) + slice][lane][0] = v;
        }
    }
    __syncthreads();

    // ---- 4. epilogue of the slot (row 0 sits in the lanes of the first 16-lane group, element 0 of the accumulator; row 1 in element
    // 1): gemv_small's, the result published as granules for the stages behind it.  Wave r of the slot finishes row r: at two rows the
    // epilogues run side by side (one wave doing both in turn cost the two-row step ~ 2 us per layer, profiles/r5q_*_b2) ---------------
    if (wslot < (NR == 8 ? 4 : R) && has_group) {
        // (the slices' sums are READ first -- all in flight -- and added afterwards in slice order: read and added one by one they were up to
        // sixteen dependent LDS round trips: the wide stage's epilogue 0.85 -> 0.66 us, profiles/r5v_*)
        constexpr int MAXS = WIDE ? 16 : 4;
        float4v part[MAXS];
#pragma unroll
        for (int w = 0; w < MAXS; ++w) part[w] = *(const float4v*)&s_red[(WIDE ? 0 : 4 * slot) + (w < slices ? w : 0)][lane][0];
        float4v sum = part[0];
#pragma unroll
        for (int w = 1; w < MAXS; ++w)
            if (w < slices) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] += part[w][r];
            }
        const float bias = has_bias ? (float)bias_raw : 0.f;
        {
            // (NR <= 4: wave-uniform.  NR == 8: rows 4-7 sit in the second 16-lane group, same accumulator element: lanes 16-31 finish row wslot + 4)
            const int r = NR == 1 ? 0 : (NR == 8 ? wslot + 4 * min(g, 1) : wslot);
            const bool live = NR == 8 ? (g < 2 && r < R) : g == 0;
            const float y = NR == 1 ? sum[0] : (NR == 2 ? (wslot == 0 ? sum[0] : sum[1]) : (wslot < 2 ? (wslot == 0 ? sum[0] : sum[1]) : (wslot == 2 ? sum[2] : sum[3])));      // row r; the lanes beyond the first 16 hold rows that do not exist
            if (st.mode == 0) {
                if (live) {
                    p.out32[r * n_out + col] = y;             // raw sums for the attention kernel of the next launch
                    unsigned long long* gq = s == p.cross_at ? p.gran_q : p.gran_s;      // ... or of this launch's attention stages: cross (q) | self (q, k, v of the next layer)
                    if (gq)
                        __hip_atomic_store((chain_gu64*)(gq + r * n_out + col), ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, y),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                const float y16 = r16(y + bias);              // the Linear's fp16 output
                h16 out;
                if (st.mode == 1) {
                    out = (h16)(p.gelu_kind == 2 ? gelu_tanh(y16) : gelu_erf(y16));
                } else {                                      // mode 2: the residual stream, this slot's 16 channels
                    const h16 xo = own_valid ? s_own[slot * NR + r][rl] : p.x[(NR == 8 ? min(r, R - 1) : r) * n_out + col];
                    out = (h16)r16((float)xo + y16);
                    if (live) { s_own[slot * NR + r][rl] = out; p.x[r * n_out + col] = out; }
                }
                // two channels per granule: the even lane stores {epoch, own | neighbour << 16}
                const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, out);
                const unsigned nb_bits = __shfl_xor(bits, 1);
                if (live && (rl & 1) == 0) {
                    unsigned long long* dst = (st.mode == 1 ? p.gran_h : p.gran_x) + r * (n_out >> 1) + (col >> 1);
                    __hip_atomic_store((chain_gu64*)dst, ((unsigned long long)epoch << 32) | (bits | (nb_bits << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if (st.mode == 2) own_valid = true;
    __syncthreads();                                          // s_red / s_in are the next stage's
}

// ---- the cross-attention of the row as the chain's last stage --------------------------------------------------------------
// attn_cross_kernel<1>'s arithmetic for ONE (head, key-range piece) per workgroup (attn_decode.hip: lane -> 8 dims of a key row,
// 8 rows per wave instruction, wave w takes rows 32 w + 128 k ..., scores in LDS, maxima and sums met in wave order, P.V per lane
// over its rows in sequence, the partial (max, sum, o[64]) left for the merge) -- bit for bit: the same expressions in the same
// order, only the rows come from LDS, where waves 4-7 put them by DMA at the START of the launch (K and V do not depend on the
// activations: their 96 KB fly while the chain's Linears run) instead of from memory through a register pipeline.
constexpr int CHAIN_CROSS_KEYS = 1536;
constexpr size_t CHAIN_DYN_LDS = 100 * 1024;      // K and V rows of a piece (beside ~ 60 KB of static LDS, of 160)
constexpr size_t CHAIN_DYN_LDS4 = 64 * 1024;      // 3 and 4 rows: the self-attention's cached rows only (beside ~ 85 KB of static LDS)
constexpr size_t CHAIN_DYN_LDS8 = 24 * 1024;      // 5 to 8 rows: the same beside ~ 128 KB (longer caches are read from memory inside the stage)
constexpr int CHAIN_MAX_LAYERS = 32;               // layers of a whole-step launch (their descriptors and tables sit in LDS; Whisper large has 32)
// what the attention stages take per LAYER (kernel arguments for a one-layer launch, the two tables in a whole-step launch)
struct ChainLayerArgs { const h16* cross_kv; const h16* cross_qbias; void* self_cache; const h16* self_bias; float self_kv_scale; };
constexpr float CHAIN_ATTN_SCALE = 0.35355339059327373f;    // 64^-0.25 (attn_decode.hip: ATTN_SCALE)
constexpr unsigned CHAIN_EPOCH_LIVE = 0x80000000u;

template <int NR>
__device__ __forceinline__ void chain_cross_prefetch(const GemvChainParams& p, const void* cross_kv, unsigned char* kv_lds, int per_split, unsigned dead) {
    // waves 4-7: this workgroup's piece of K, then of V, as 1 KiB pieces (8 rows of 128 B) into LDS, linear
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_row = p.cross_heads * p.cross_nsplit, n_items = NR * per_row;
    const int item = (int)blockIdx.x - ((int)gridDim.x - n_items);       // the LAST workgroups (see the kernel): item = (row, piece, head)
    if (wid < 4 || item < 0) return;
    const int urow = NR == 1 ? 0 : item / per_row, rem = NR == 1 ? item : item - urow * per_row;
    const int h = rem % p.cross_heads, sp = rem / p.cross_heads;
    const int k_begin = sp * per_split, nkeys = max(0, min(p.cross_Tk, k_begin + per_split) - k_begin);
    if (nkeys == 0 || ((dead >> urow) & 1)) return;      // (a finished row's K / V stay where they are)
    const int n_pieces = (nkeys + 7) >> 3;
    for (int m = 0; m < 2; ++m) {
        const unsigned char* src = (const unsigned char*)cross_kv + (size_t)urow * p.cross_row_bytes + ((size_t)(m * p.cross_heads + h) * p.cross_Tk + k_begin) * 128;
        for (int pc = wid - 4; pc < n_pieces; pc += 4) {
            const int row = min(pc * 8 + (lane >> 3), nkeys - 1);        // rows past the piece re-read its last row (never used)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)row * 128 + (lane & 7) * 16),
                                             (__attribute__((address_space(3))) void*)(kv_lds + m * per_split * 128 + pc * 1024), 16, 0, 0);
        }
    }
}

template <int NR>
__device__ __forceinline__ void chain_cross_stage(const GemvChainParams& p, const ChainLayerArgs& la, unsigned epoch_q, const unsigned char* kv_lds, int per_split,
                                                  float* s_sc, float (*s_redc)[2] /* [8] */, float (*s_o)[64], float* s_q, unsigned dead) {
    constexpr int DPL = 8, LPR = 8, RPI = 8, UNR = 4;
    constexpr int STRIDE = 4 * RPI * UNR;
    constexpr int KB = 3, KB2 = 2;                 // iterations whose rows are requested together (P.V | scores, a wave's share)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_row = p.cross_heads * p.cross_nsplit, n_items = NR * per_row;
    const int item = (int)blockIdx.x - ((int)gridDim.x - n_items);
    const bool has_item = item >= 0;
    const int urow = (has_item && NR > 1) ? item / per_row : 0, rem = item - urow * per_row;
    const int h = has_item ? rem % p.cross_heads : 0, sp = has_item ? rem / p.cross_heads : 0;
    const bool dead_item = has_item && ((dead >> urow) & 1);                                       // a finished row: nothing read, zeros published
    const int k_begin = sp * per_split, nkeys = (has_item && !dead_item) ? max(0, min(p.cross_Tk, k_begin + per_split) - k_begin) : 0;
    const unsigned long long* gran_q = p.gran_q + (size_t)urow * p.cross_heads * 64;              // this row's q sums
    unsigned long long* gran_p = p.gran_p + (size_t)urow * p.cross_heads * 66 * 4;               // ... and its pieces' partial results
    const int sub = lane % LPR, rowi = lane / LPR;
    const int first = (wid & 3) * (RPI * UNR);
    const bool worker = wid < 4 && has_item && nkeys > 0;
    half8v qb8 = half8v{0, 0, 0, 0, 0, 0, 0, 0};                              // the lane's 8 q-bias values: requested now, not behind the wait for q
    {                                                                      // (unconditional: no wait inside a branch)
        const bool hb = has_item && nkeys > 0 && la.cross_qbias != nullptr;
        const half8v raw = *(const half8v*)(hb ? la.cross_qbias + h * 64 + sub * DPL : (const h16*)p.st);
#pragma unroll
        for (int e = 0; e < DPL; ++e) qb8[e] = hb ? raw[e] : (h16)0.f;
    }
    if (wid >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's K / V pieces have landed
    if (wid == 0 && has_item) {                                               // the head's 64 q sums, as the last Linear published them (a finished row waits
                                                                              // for them too: its zeros must not replace the last layer's pieces before those are read)
        int fst[1] = {h * 64 + 2 * min(lane, 31)};
        u32x4 val[1];
        if (sweep_granules16<1>(gran_q, fst, epoch_q, val, p.err, lane) && lane < 32) {
            unsigned q0 = val[0].x, q1 = val[0].z;
            asm volatile("" : "+v"(q0), "+v"(q1));            // (kept apart: the pair was stored as (x, x) when the compiler formed it itself)
            s_q[2 * lane] = __builtin_bit_cast(float, q0);
            s_q[2 * lane + 1] = __builtin_bit_cast(float, q1);
        }
    }
    __syncthreads();                                                           // (A) rows and q sums are in LDS
    // the piece's partial result (max, sum, o[64]) as tagged granules, [head][66][4 pieces]: a lane of the merge finds the four
    // pieces of a value side by side
    auto put = [&](int r, float v) {
        __hip_atomic_store((chain_gu64*)(gran_p + ((size_t)h * 66 + r) * 4 + sp), ((unsigned long long)epoch_q << 32) | __builtin_bit_cast(unsigned, v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (has_item && nkeys == 0) {                                              // an empty piece: the neutral element (attn_cross_kernel)
        // (a finished row: maximum 0, sum 1 in piece 0 and 0 elsewhere, o = 0 -- the merge then yields a FINITE zero row)
        if (tid < 66) put(tid, dead_item ? ((tid == 1 && sp == 0) ? 1.f : 0.f) : ((tid == 0) ? -INFINITY : 0.f));
    }
    const unsigned char* K = kv_lds;
    const unsigned char* V = kv_lds + (size_t)per_split * 128;
    const int nb = (nkeys + STRIDE - 1) / STRIDE;
    float qf[DPL];
    float mx = -INFINITY;
    if (has_item && nkeys > 0) {
        // q of this lane's 8 dims: one slab, bias, the two roundings (attn_cross_kernel's prologue at ksplit = 1)
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const float x = s_q[sub * DPL + e];
            float qa = 0.f;
            qa += x + 0.f;
            qa += 0.f + 0.f;
            const float bs = (float)qb8[e];
            qf[e] = r16(r16(qa + bs) * CHAIN_ATTN_SCALE);
        }
        // ---- pass 1: scores -- a key's score is its own (no sum across keys), so ALL EIGHT waves take part: wave w scores the rows
        // attn_cross_kernel's wave w & 3 scores, in the iterations k with k & 1 == w >> 2 (the upper four waves have nothing else to
        // do once their K / V rows have landed).  The rows of two iterations are read from LDS before the first is used. ----------
        // (tried in round 5: units of 8 rows dealt evenly over the eight waves -- six each instead of eight on waves 0-3 and four on waves
        // 4-7 -- with a wave's six rows read together: the pass took 1.50 instead of 1.36 us, profiles/r5v_*; not kept)
        const int half = wid >> 2;
        for (int k0 = half; k0 < nb; k0 += 2 * KB2) {
            half8v hv[KB2][UNR];
#pragma unroll
            for (int kk = 0; kk < KB2; ++kk)
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int rr = min(first + (k0 + 2 * kk) * STRIDE + u * RPI + rowi, nkeys - 1);
                    hv[kk][u] = *(const half8v*)(K + (size_t)rr * 128 + sub * 16);
                }
#pragma unroll
            for (int kk = 0; kk < KB2; ++kk) {
                if (k0 + 2 * kk >= nb) break;                                  // wave-uniform
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int r = first + (k0 + 2 * kk) * STRIDE + u * RPI + rowi;
                    float ks[DPL];
#pragma unroll
                    for (int e = 0; e < DPL; ++e) ks[e] = r16((float)hv[kk][u][e] * CHAIN_ATTN_SCALE);
                    float acc = 0.f;
#pragma unroll
                    for (int e = 0; e < DPL; ++e) acc = fmaf(qf[e], ks[e], acc);
                    acc += wave_dpp<0xB1>(acc);
                    acc += wave_dpp<0x4E>(acc);
                    acc += wave_dpp<0x141>(acc);
                    const float sc = r16(f32_as_is(acc));
                    if (r < nkeys) {
                        if (sub == 0) s_sc[r] = sc;
                        mx = fmaxf(mx, sc);
                    }
                }
            }
        }
        const float m = wave_max_nomfma(mx);
        if (lane == 0) s_redc[wid][0] = m;
    }
    __syncthreads();                                                           // (B)
    float gmax = 0.f, gsum = 0.f;
    if (worker) {
        gmax = fmaxf(fmaxf(fmaxf(s_redc[0][0], s_redc[1][0]), fmaxf(s_redc[2][0], s_redc[3][0])),
                     fmaxf(fmaxf(s_redc[4][0], s_redc[5][0]), fmaxf(s_redc[6][0], s_redc[7][0])));      // (a maximum: whoever found it)
        float sm = 0.f;
        for (int j = tid; j < nkeys; j += 256) {
            const float e = __expf(s_sc[j] - gmax);
            s_sc[j] = e;
            sm += e;
        }
        sm = wave_sum_nomfma(sm);
        if (lane == 0) s_redc[wid][1] = sm;
    }
    __syncthreads();                                                           // (C)
    if (worker) {
        gsum = s_redc[0][1] + s_redc[1][1] + s_redc[2][1] + s_redc[3][1];
        // ---- pass 2: P.V (probabilities unnormalised: the merge divides) ---------------------------------------------------
        float o[DPL];
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] = 0.f;
        for (int k0 = 0; k0 < nb; k0 += KB) {                              // (rows and weights of three iterations requested together, added in the old order)
            half8v hv[KB][UNR];
            float pr[KB][UNR];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int r = first + (k0 + kk) * STRIDE + u * RPI + rowi;
                    hv[kk][u] = *(const half8v*)(V + (size_t)min(r, nkeys - 1) * 128 + sub * 16);
                    pr[kk][u] = r < nkeys ? s_sc[min(r, nkeys - 1)] : 0.f;
                }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                if (k0 + kk >= nb) break;                                      // wave-uniform
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
#pragma unroll
                    for (int e = 0; e < DPL; ++e) o[e] = fmaf(pr[kk][u], (float)hv[kk][u][e], o[e]);
                }
#pragma unroll
                for (int e = 0; e < DPL; ++e) asm volatile("" : "+v"(o[e]) : : "memory");
            }
        }
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] += wave_dpp<0x128>(o[e]);
        wave_add_xor16_x8_nomfma(o);                               // (attn_cross_kernel's steps, eight values at a time: no matrix instruction
        wave_add_xor32_x8_nomfma(o);                               // of this wave is in flight here -- its last Linear stage ended microseconds ago)
        if (rowi == 0) {
#pragma unroll
            for (int e = 0; e < DPL; ++e) s_o[wid][sub * DPL + e] = o[e];
        }
    }
    __syncthreads();                                                           // (D)
    if (worker && tid < 64) {
        const float v = s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid];
        put(2 + tid, v);
        if (tid == 0) { put(0, gmax); put(1, gsum); }
    }
}

// The cross-attention stage of the 3- and 4-row step.  R x heads x 4 pieces are more items than the launch has workgroups, and four rows' K / V
// (30.7 MB per layer) more than the LDS beside the Linears' buffers holds, so here an item is attn_cross_kernel<1>'s own shape -- FOUR waves --
// two items per workgroup (waves 0-3 | 4-7; the last ceil(items / 2) workgroups of the launch), and the rows come from memory into
// REGISTERS: a wave's share of a piece is 12 K and 12 V loads of 16 bytes per lane, all 24 requested at the head of the stage -- the
// workgroups that carry items idle through the two Linear stages in front of it (out, cq), so the rows fly while q is still being made.
// The arithmetic is attn_cross_kernel's, bit for bit: wave w of the item scores rows 32 w + 128 k ..., maxima and sums met in wave order,
// P.V per lane over its rows in sequence, (max, sum, o[64]) left for the merge as tagged granules.
__device__ __forceinline__ void chain_cross_stage4(const GemvChainParams& p, const ChainLayerArgs& la, unsigned epoch_q, int per_split,
                                                   float* s_sc_all /* [2][768] */, float (*s_redc)[2] /* [8] */, float (*s_o)[64] /* [8] */, float (*s_q)[64] /* [2] */, unsigned dead) {
    constexpr int DPL = 8, LPR = 8, RPI = 8, UNR = 4;
    constexpr int STRIDE = 4 * RPI * UNR;          // 128 rows per iteration of an item's four waves
    constexpr int KB = 3;                          // iterations of a piece (<= 384 keys)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wid >> 2, wq = wid & 3, tl = tid & 255;
    const int per_row = p.cross_heads * p.cross_nsplit, n_items = p.rows * per_row;
    const int rounds = (n_items + 511) >> 9;       // items beyond two per workgroup of the launch: a second round (5-8 rows of large-v2: 7, 8)
    const int n_wgs = (n_items + 2 * rounds - 1) / (2 * rounds);
    const int wg = (int)blockIdx.x - ((int)gridDim.x - n_wgs);
    if (wg < 0) {                                  // (workgroup-uniform) no items here: the stage's barriers, nothing requested
        for (int b = 0; b < 5 * rounds - 1; ++b) __syncthreads();
        return;
    }
    for (int rd = 0; rd < rounds; ++rd) {
    const int item = 2 * (rd * n_wgs + wg) + half;
    const bool has_item = item < n_items;
    const int urow = has_item ? item / per_row : 0, rem = has_item ? item - urow * per_row : 0;
    const int h = rem % p.cross_heads, sp = rem / p.cross_heads;
    const bool dead_item = has_item && ((dead >> urow) & 1);                   // a finished row: nothing read, zeros published
    const int k_begin = sp * per_split, nkeys = (has_item && !dead_item) ? max(0, min(p.cross_Tk, k_begin + per_split) - k_begin) : 0;
    const bool worker = has_item && nkeys > 0;
    float* s_sc = s_sc_all + half * 768;
    const unsigned long long* gran_q = p.gran_q + (size_t)urow * p.cross_heads * 64;
    unsigned long long* gran_p = p.gran_p + (size_t)urow * p.cross_heads * 66 * 4;
    const int sub = lane % LPR, rowi = lane / LPR;
    const int first = wq * (RPI * UNR);
    // ---- every row of this wave, K and V, requested now ------------------------------------------------------------------------------
    u32x4 kv_k[KB][UNR], kv_v[KB][UNR];
    {
        const unsigned char* Kg = (const unsigned char*)la.cross_kv + (size_t)urow * p.cross_row_bytes + ((size_t)(0 * p.cross_heads + h) * p.cross_Tk + k_begin) * 128;
        const unsigned char* Vg = (const unsigned char*)la.cross_kv + (size_t)urow * p.cross_row_bytes + ((size_t)(1 * p.cross_heads + h) * p.cross_Tk + k_begin) * 128;
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int rr = max(min(first + kk * STRIDE + u * RPI + rowi, nkeys - 1), 0);      // (no item: row 0 of item 0's K -- valid memory, never used)
                kv_k[kk][u] = __builtin_nontemporal_load((const u32x4*)(Kg + (size_t)rr * 128 + sub * 16));
            }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int rr = max(min(first + kk * STRIDE + u * RPI + rowi, nkeys - 1), 0);
                kv_v[kk][u] = __builtin_nontemporal_load((const u32x4*)(Vg + (size_t)rr * 128 + sub * 16));
            }
    }
    half8v qb8 = half8v{0, 0, 0, 0, 0, 0, 0, 0};
    {
        const bool hb = worker && la.cross_qbias != nullptr;
        const half8v raw = *(const half8v*)(hb ? la.cross_qbias + h * 64 + sub * DPL : (const h16*)p.st);
#pragma unroll
        for (int e = 0; e < DPL; ++e) qb8[e] = hb ? raw[e] : (h16)0.f;
    }
    if (wq == 0 && has_item) {                                                  // the head's 64 q sums, as the last Linear published them (a finished row waits too)
        int fst[1] = {h * 64 + 2 * min(lane, 31)};
        u32x4 val[1];
        if (sweep_granules16<1>(gran_q, fst, epoch_q, val, p.err, lane) && lane < 32) {
            unsigned q0 = val[0].x, q1 = val[0].z;
            asm volatile("" : "+v"(q0), "+v"(q1));
            s_q[half][2 * lane] = __builtin_bit_cast(float, q0);
            s_q[half][2 * lane + 1] = __builtin_bit_cast(float, q1);
        }
    }
    __syncthreads();                                                           // (A) the q sums are in LDS
    auto put = [&](int r, float v) {
        __hip_atomic_store((chain_gu64*)(gran_p + ((size_t)h * 66 + r) * 4 + sp), ((unsigned long long)epoch_q << 32) | __builtin_bit_cast(unsigned, v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (has_item && nkeys == 0) {                  // an empty piece: the neutral element; a finished row: maximum 0, sum 1 in piece 0, o = 0 (a finite zero row)
        if (tl < 66) put(tl, dead_item ? ((tl == 1 && sp == 0) ? 1.f : 0.f) : ((tl == 0) ? -INFINITY : 0.f));
    }
    const int nb = (nkeys + STRIDE - 1) / STRIDE;
    float qf[DPL];
    float mx = -INFINITY;
    if (worker) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const float x = s_q[half][sub * DPL + e];
            float qa = 0.f;
            qa += x + 0.f;
            qa += 0.f + 0.f;
            const float bs = (float)qb8[e];
            qf[e] = r16(r16(qa + bs) * CHAIN_ATTN_SCALE);
        }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            if (kk >= nb) break;                                               // wave-uniform
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = first + kk * STRIDE + u * RPI + rowi;
                const half8v hv = __builtin_bit_cast(half8v, kv_k[kk][u]);
                float ks[DPL];
#pragma unroll
                for (int e = 0; e < DPL; ++e) ks[e] = r16((float)hv[e] * CHAIN_ATTN_SCALE);
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < DPL; ++e) acc = fmaf(qf[e], ks[e], acc);
                acc += wave_dpp<0xB1>(acc);
                acc += wave_dpp<0x4E>(acc);
                acc += wave_dpp<0x141>(acc);
                const float sc = r16(f32_as_is(acc));
                if (r < nkeys) {
                    if (sub == 0) s_sc[r] = sc;
                    mx = fmaxf(mx, sc);
                }
            }
        }
        const float m = wave_max_nomfma(mx);
        if (lane == 0) s_redc[wid][0] = m;
    }
    __syncthreads();                                                           // (B)
    float gmax = 0.f, gsum = 0.f;
    if (worker) {
        gmax = fmaxf(fmaxf(s_redc[4 * half][0], s_redc[4 * half + 1][0]), fmaxf(s_redc[4 * half + 2][0], s_redc[4 * half + 3][0]));
        float sm = 0.f;
        for (int j = tl; j < nkeys; j += 256) {
            const float e = __expf(s_sc[j] - gmax);
            s_sc[j] = e;
            sm += e;
        }
        sm = wave_sum_nomfma(sm);
        if (lane == 0) s_redc[wid][1] = sm;
    }
    __syncthreads();                                                           // (C)
    if (worker) {
        gsum = s_redc[4 * half][1] + s_redc[4 * half + 1][1] + s_redc[4 * half + 2][1] + s_redc[4 * half + 3][1];
        float o[DPL];
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] = 0.f;
        float pr[KB][UNR];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = first + kk * STRIDE + u * RPI + rowi;
                pr[kk][u] = r < nkeys ? s_sc[min(r, nkeys - 1)] : 0.f;
            }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            if (kk >= nb) break;                                               // wave-uniform
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const half8v hv = __builtin_bit_cast(half8v, kv_v[kk][u]);
#pragma unroll
                for (int e = 0; e < DPL; ++e) o[e] = fmaf(pr[kk][u], (float)hv[e], o[e]);
            }
#pragma unroll
            for (int e = 0; e < DPL; ++e) asm volatile("" : "+v"(o[e]) : : "memory");
        }
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[e] += wave_dpp<0x128>(o[e]);
        wave_add_xor16_x8_nomfma(o);
        wave_add_xor32_x8_nomfma(o);
        if (rowi == 0) {
#pragma unroll
            for (int e = 0; e < DPL; ++e) s_o[wid][sub * DPL + e] = o[e];
        }
    }
    __syncthreads();                                                           // (D)
    if (worker && tl < 64) {
        const float v = s_o[4 * half][tl] + s_o[4 * half + 1][tl] + s_o[4 * half + 2][tl] + s_o[4 * half + 3][tl];
        put(2 + tl, v);
        if (tl == 0) { put(0, gmax); put(1, gsum); }
    }
    if (rd + 1 < rounds) __syncthreads();                                      // (E) s_sc / s_o / s_q are the next round's
    }
}

// The merge of the cross-attention's four key-range pieces when they were produced by THIS launch (tagged granules, p.gran_p):
// attn_cross_combine_kernel's arithmetic -- m = max of the pieces' maxima, f_q = exp(m_q - m), den = sum_q l_q f_q and
// num = sum_q o_q f_q in piece order, (h16)(num / den) -- with every lane holding all four (m, l) pairs itself instead of taking
// them from lanes 0-3 (the same values: fmaxf and the products do not depend on who computes them).  All eight waves share the
// heads (three per wave at 20 heads), a wave's 18 loads are in flight together; the merged row goes to s_in[0].  Ends with a barrier.
template <int NR>
__device__ __forceinline__ void chain_merge_tagged(const GemvChainParams& p, unsigned tag, h16 (*s_in)[CHAIN_MAX_IN + 8]) {
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The (row, head) pairs -- pair = row * heads + head: the pieces' granules of the rows lie back to back -- are dealt over the eight waves,
    // MH per wave and pass: three at one row (20 heads), five at two rows (40 pairs): ONE pass either way (the rows one after the other cost
    // the two-row step a second pass, ~ 1.2 us per layer).  Per pair a lane loads its own output value's four pieces (2 x 16 B) and ONE 16-byte
    // chunk of the pair's eight (maximum, sum) values -- lanes 0-3 the four chunks, the others repeat them -- which are the same for every
    // lane and read across the wave (v_readlane): 3 loads per pair instead of 6 (15 per pass at two rows; 30 spilled 164 bytes per lane).
    constexpr int MH = NR == 1 ? 3 : 5;
    const int n_pairs = (NR == 1 ? 1 : p.rows) * p.merge_heads;      // (NR == 4: 60 or 80 pairs, two passes)
    for (int i0 = wid; i0 < n_pairs; i0 += 8 * MH) {
        int first[3 * MH];
        u32x4 val[3 * MH];
#pragma unroll
        for (int u = 0; u < MH; ++u) {
            const int pr = min(i0 + 8 * u, n_pairs - 1);
            first[3 * u + 0] = (pr * 66) * 4 + 2 * (lane & 3);       // granules 0-7 of the pair: m[4] | l[4], chunk lane & 3
            first[3 * u + 1] = (pr * 66 + 2 + lane) * 4; first[3 * u + 2] = (pr * 66 + 2 + lane) * 4 + 2;
        }
        if (!sweep_granules16<3 * MH>(p.gran_p, first, tag, val, p.err, lane)) break;
#pragma unroll
        for (int u = 0; u < MH; ++u) {
            const int pr = i0 + 8 * u;
            if (pr >= n_pairs) break;                                 // wave-uniform
            const int row = NR == 1 ? 0 : pr / p.merge_heads, h = pr - row * p.merge_heads;
            unsigned cx = val[3 * u].x, cz = val[3 * u].z;
            asm volatile("" : "+v"(cx), "+v"(cz));
            unsigned mb[4], lb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                             // piece q: chunk q >> 1 (maxima) / 2 + (q >> 1) (sums), its first or second granule
                mb[q] = (unsigned)__builtin_amdgcn_readlane((int)((q & 1) ? cz : cx), q >> 1);
                lb[q] = (unsigned)__builtin_amdgcn_readlane((int)((q & 1) ? cz : cx), 2 + (q >> 1));
            }
            unsigned ob[4] = {val[3 * u + 1].x, val[3 * u + 1].z, val[3 * u + 2].x, val[3 * u + 2].z};
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(ob[q]));      // (see chain_cross_stage: pairs formed by the compiler from a granule load came out as (x, x))
            float mq[4], lq[4], oq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { mq[q] = __builtin_bit_cast(float, mb[q]); lq[q] = __builtin_bit_cast(float, lb[q]); oq[q] = __builtin_bit_cast(float, ob[q]); }
            const float m = fmaxf(fmaxf(mq[0], mq[1]), fmaxf(mq[2], mq[3]));
            float den = 0.f, num = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float f = __expf(mq[q] - m);
                den += mul_rn(lq[q], f);                              // (products of their own, as attn_cross_combine_kernel forms them: common.h)
                num += mul_rn(oq[q], f);
            }
            s_in[row][h * 64 + lane] = (h16)(num / den);
        }
    }
    __syncthreads();
}

// The cached K and V rows of a self-attention head into LDS, a layer AHEAD of their use (waves 4-7 of the head's workgroup: they idle
// through every other stage, and the dynamic LDS the cross-attention's workgroups fill with K/V rows is free here).  Rows 0 .. T - 1
// are contiguous in the cache, so each matrix is copied as 1 KiB pieces, linear: the stage reads them where it read global memory.
// Returns the bytes per matrix (the V rows start there), or 0 when there is nothing to fetch or the rows do not fit.
template <bool I8>
__device__ __forceinline__ int chain_self_prefetch(const GemvChainParams& p, const void* cache, int T, int h, unsigned char* kv_lds, int avail) {
    constexpr int ROW_B = I8 ? 64 : 128;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_pieces = (T * ROW_B + 1023) >> 10, total = (p.self_cap * ROW_B) >> 10;        // pieces wanted | pieces the head's matrix has
    if (T <= 0 || 2 * n_pieces * 1024 > avail || n_pieces > total) return 0;
    if (wid >= 4) {
        for (int m = 0; m < 2; ++m) {
            const unsigned char* src = (const unsigned char*)cache + ((size_t)(m * p.self_heads + h) * p.self_cap * 64) * (I8 ? 1 : 2);
            for (int pc = wid - 4; pc < n_pieces; pc += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)pc * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(kv_lds + (m * n_pieces + pc) * 1024), 16, 0, 0);
        }
    }
    return n_pieces * 1024;
}

// ---- the self-attention of the row as the launch's first stage -----------------------------------------------------------------
// attn_self_wg_kernel (attn_decode.hip) at one new token of one utterance, for ONE head per workgroup: its four waves are this
// workgroup's waves 0-3, same expressions in the same order (q / k / v = the qkv sums of the launch before + bias, rounded; the
// cache append at position T; key blocks of 64 dealt over the waves, a key per lane; the two-pass softmax with fp16 probabilities;
// P.V by wave-wide 16-byte loads of whole V rows, partial sums added in (wave, row, block) order) -- bit for bit.  The head's 64
// outputs are published as granules (p.gran_c, tagged with the launch's epoch) for the out projection, this launch's next stage.
template <bool I8>
__device__ __forceinline__ void chain_self_stage(const GemvChainParams& p, const ChainLayerArgs& la_, int T, unsigned epoch0, unsigned tag_s, int h, int urow, float* s_p, h16 (*s_new)[64],
                                                 float (*s_r2)[4], float* s_o_flat, float* s_lut, const unsigned char* lds_rows, int lds_v_off) {
    ChainLayerArgs la = la_;                                  // this row's share of the cache ([row][2][H][cap][64])
    la.self_cache = (unsigned char*)la_.self_cache + (size_t)urow * p.self_row_bytes;
    constexpr float SCALE = 0.35355339059327373f;     // 64^-0.25 (attn_decode.hip: ATTN_SCALE)
    constexpr int ES = I8 ? 1 : 2;
    constexpr int ROW_B = 64 * ES;
    constexpr int DIMS = 16 / ES;
    constexpr int NCH = 64 / DIMS;
    constexpr int VROWS = 64 / NCH;
    constexpr int KCH = ROW_B / 16;
    constexpr int VPRE = 4;
    constexpr int NW = 4, NT = 64 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool worker = wid < NW;
    h16* s_q = s_new[0]; h16* s_knew = s_new[1]; h16* s_vnew = s_new[2];
    const int H = p.self_heads, C = H * 64;
    // the cached rows: in LDS already (chain_self_prefetch, a layer ago; lds_v_off > 0), or from memory, the first blocks requested now
    const bool in_lds = lds_v_off > 0;                                  // (workgroup-uniform)
    const unsigned char* pastK = in_lds ? lds_rows : (const unsigned char*)la.self_cache + ((size_t)(0 * H + h) * p.self_cap * 64) * ES;
    const unsigned char* pastV = in_lds ? lds_rows + lds_v_off : (const unsigned char*)la.self_cache + ((size_t)(1 * H + h) * p.self_cap * 64) * ES;
    const int vr = lane / NCH, vc = lane % NCH;
    // the first K block and the first four V blocks of this wave into registers, all requests in flight together: from memory at the
    // stage's start, or -- rows that chain_self_prefetch has put into LDS -- behind the stage's first barrier, when they have landed
    uint4 kpre[KCH], vpre[VPRE];
    auto rows_to_registers = [&]() {
        const int kr = min(64 * wid + lane, T - 1);
#pragma unroll
        for (int c = 0; c < KCH; ++c) kpre[c] = ((const uint4*)(pastK + (size_t)kr * ROW_B))[c];
#pragma unroll
        for (int n = 0; n < VPRE; ++n) {
            const int row = min((wid + NW * n) * VROWS + vr, T - 1);
            vpre[n] = *(const uint4*)(pastV + (size_t)row * ROW_B + vc * 16);
        }
    };
    if (worker && T > 0 && !in_lds) rows_to_registers();
    const float t_dq = la.self_kv_scale;
    const float inv_t = 1.0f / la.self_kv_scale;
    // int8 cache: what a cached CODE contributes depends on the code and the layer's scale only -- r16(r16(code * t) * SCALE) to a score,
    // r16(code * t) to P.V (attn_self_wg_kernel's expressions) -- so the 2 x 256 values are computed once per stage (every thread one)
    // and looked up: the same bits, an LDS read instead of seven vector instructions per cached element (64 dims x 11 instructions per
    // key and lane were ~ 1.4 us of the stage; round 5)
    if constexpr (I8) {
        const float d = r16((float)(int)(int8_t)(tid & 255) * t_dq);
        s_lut[tid] = tid < 256 ? r16(d * SCALE) : d;
    }
    float k_new = 0.f, v_new = 0.f;
    if (wid == 0) {                                   // this call's q, k, v of the head (lane = dim): one slab, bias, fp16
        float q = 0.f, k = 0.f, v = 0.f;
        // (requested ahead of the wait for the sums, as UNCONDITIONAL loads: behind `bias ? ... : 0` each is waited for in its branch)
        const h16* bsrc = la.self_bias ? la.self_bias + h * 64 + lane : (const h16*)p.st;
        const int bstep = la.self_bias ? C : 0;
        const h16 bq_raw = bsrc[0], bk_raw = bsrc[bstep], bv_raw = bsrc[2 * bstep];
        if (p.gran_s) {                               // the sums the qkv stage of THIS launch has just published (the whole step in one launch)
            const int r0 = urow * 3 * C;              // (rows of 3 C sums)
            int first[3] = {r0 + h * 64 + 2 * min(lane, 31), r0 + C + h * 64 + 2 * min(lane, 31), r0 + 2 * C + h * 64 + 2 * min(lane, 31)};
            u32x4 val[3];
            float* s_f = s_p;                         // (s_p is not in use yet: 3 x 64 floats)
            if (sweep_granules16<3>(p.gran_s, first, tag_s, val, p.err, lane) && lane < 32) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    unsigned a0 = val[m].x, a1 = val[m].z;
                    asm volatile("" : "+v"(a0), "+v"(a1));        // (see chain_cross_stage)
                    s_f[m * 64 + 2 * lane] = __builtin_bit_cast(float, a0);
                    s_f[m * 64 + 2 * lane + 1] = __builtin_bit_cast(float, a1);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's LDS writes before its reads
            q += s_f[lane]; k += s_f[64 + lane]; v += s_f[128 + lane];
        } else {
            const float* part = p.self_part + (size_t)urow * 3 * C + h * 64 + lane;
            q += part[0]; k += part[C]; v += part[2 * C];
        }
        q = r16(q + (la.self_bias ? (float)bq_raw : 0.f));
        k = r16(k + (la.self_bias ? (float)bk_raw : 0.f));
        v = r16(v + (la.self_bias ? (float)bv_raw : 0.f));
        s_knew[lane] = (h16)k;
        s_vnew[lane] = (h16)v;
        if constexpr (I8) s_lut[512 + lane] = r16(k * SCALE);      // the new key's 64 score factors (k is an fp16 value already)
        k_new = k; v_new = v;                         // (the cache append waits for the end of the stage: see there)
        s_q[lane] = (h16)r16(q * SCALE);
    }
    if (wid >= 4 && in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the waves that requested the rows: they have landed
    __syncthreads();
    if (worker && T > 0 && in_lds) rows_to_registers();
    const int nk = T + 1;
    float mx = -INFINITY;
    if (worker) {
        // (q stays in LDS: one broadcast read per product.  The head's 64 q values in registers, read once, made the score pass 1.8 ->
        // 2.75 us -- profiles/r5t_*)
        for (int kb = wid; kb * 64 < nk; kb += NW) {
            const int j = kb * 64 + lane;
            float sc = -INFINITY;
            if (j < nk) {
                // (tried in round 5: ONE chain for cached and new keys, the new key's lane selecting its factors per element instead of
                // running a branch of its own -- faster in attn_self_wg_kernel, 0.5 us SLOWER here (profiles/r5s_*): kept as two branches)
                float acc = 0.f;
                if (j < T) {
                    const uint4* kr = (const uint4*)(pastK + (size_t)j * ROW_B);
#pragma unroll
                    for (int c = 0; c < KCH; ++c) {
                        const uint4 w = kb == wid ? kpre[c] : kr[c];
                        if (I8) {
                            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const float kd = s_lut[(ws[e >> 2] >> (8 * (e & 3))) & 0xff];      // = r16(r16((float)code * t_dq) * SCALE)
                                acc = fmaf((float)s_q[c * 16 + e], kd, acc);
                            }
                        } else {
                            const half8v wh = __builtin_bit_cast(half8v, w);
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc = fmaf((float)s_q[c * 8 + e], r16((float)wh[e] * SCALE), acc);
                        }
                    }
                } else {
                    if constexpr (I8) {                       // the new key: its 64 factors r16(k * SCALE) wait behind the tables
#pragma unroll 8
                        for (int e = 0; e < 64; ++e) acc = fmaf((float)s_q[e], s_lut[512 + e], acc);
                    } else {
                        const h16* kn = s_knew;
#pragma unroll 8
                        for (int e = 0; e < 64; ++e) acc = fmaf((float)s_q[e], r16((float)kn[e] * SCALE), acc);
                    }
                }
                sc = r16(f32_as_is(acc));
                s_p[j] = sc;
            }
            mx = fmaxf(mx, sc);
        }
        mx = wave_max_nomfma(mx);
        if (lane == 0) s_r2[0][wid] = mx;
    }
    __syncthreads();
    float e0 = 0.f, e1 = 0.f;
    if (worker) {
        mx = fmaxf(fmaxf(s_r2[0][0], s_r2[0][1]), fmaxf(s_r2[0][2], s_r2[0][3]));
        e0 = tid < nk ? __expf(s_p[tid] - mx) : 0.f;
        e1 = tid + NT < nk ? __expf(s_p[tid + NT] - mx) : 0.f;
        const float wsum = wave_sum_nomfma(e0 + e1);
        if (lane == 0) s_r2[1][wid] = wsum;
    }
    __syncthreads();
    if (worker) {
        const float inv = 1.0f / ((s_r2[1][0] + s_r2[1][1]) + (s_r2[1][2] + s_r2[1][3]));
        if (tid < nk) s_p[tid] = r16(e0 * inv);
        if (tid + NT < nk) s_p[tid + NT] = r16(e1 * inv);
    }
    __syncthreads();
    if (worker) {
        float o[DIMS];
#pragma unroll
        for (int d = 0; d < DIMS; ++d) o[d] = 0.f;
        auto add_block = [&](int vb, const uint4& w) {
            const int row = vb * VROWS + vr;
            const float pj = row < T ? s_p[row] : 0.f;
            if (I8) {
                const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int d = 0; d < 16; ++d)
                    o[d] = fmaf(pj, s_lut[256 + ((ws[d >> 2] >> (8 * (d & 3))) & 0xff)], o[d]);      // = r16((float)code * t_dq)
            } else {
                const half8v wh = __builtin_bit_cast(half8v, w);
#pragma unroll
                for (int d = 0; d < 8; ++d) o[d] = fmaf(pj, (float)wh[d], o[d]);
            }
        };
#pragma unroll
        for (int n = 0; n < VPRE; ++n) {
            const int vb = wid + NW * n;
            if (vb * VROWS < T) add_block(vb, vpre[n]);
        }
        for (int vb = wid + NW * VPRE; vb * VROWS < T; vb += NW) {
            const int row = min(vb * VROWS + vr, T - 1);
            add_block(vb, *(const uint4*)(pastV + (size_t)row * ROW_B + vc * 16));
        }
#pragma unroll
        for (int d = 0; d < DIMS; ++d) s_o_flat[tid * (DIMS + 1) + d] = o[d];
    }
    __syncthreads();
    if (wid == 0) {
        const int ch = lane / DIMS, d = lane % DIMS;
        // (the NW x VROWS partial sums are READ first, all of them in flight, and added afterwards in attn_self_wg_kernel's (wave, row) order:
        // read and added one by one they were 64 dependent LDS round trips, most of the stage's last 1.2 us)
        float part[NW][VROWS];
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int r = 0; r < VROWS; ++r) part[w][r] = s_o_flat[(w * 64 + r * NCH + ch) * (DIMS + 1) + d];
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int r = 0; r < VROWS; ++r) acc += part[w][r];
        for (int j = T; j < nk; ++j) acc = fmaf(s_p[j], (float)s_vnew[lane], acc);
        const h16 out = (h16)f32_as_is(acc);
        if (p.self_out) p.self_out[urow * C + h * 64 + lane] = out;      // (plain copy: tests)
        const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, out);
        const unsigned nb_bits = __shfl_xor(bits, 1);
        if ((lane & 1) == 0)
            __hip_atomic_store((chain_gu64*)(p.gran_c + ((urow * C + h * 64 + lane) >> 1)), ((unsigned long long)epoch0 << 32) | (bits | (nb_bits << 16)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The cache append, position T: LAST.  A workgroup barrier waits for the wave's stores to be acknowledged by memory (2 us here),
        // and nothing in this launch reads the new row from memory -- the next token step does.
        const size_t off_k = ((size_t)(0 * H + h) * p.self_cap + T) * 64 + lane;
        const size_t off_v = ((size_t)(1 * H + h) * p.self_cap + T) * 64 + lane;
        if (I8) {
            ((int8_t*)la.self_cache)[off_k] = (int8_t)fminf(127.f, fmaxf(-128.f, rintf(k_new * inv_t)));
            ((int8_t*)la.self_cache)[off_v] = (int8_t)fminf(127.f, fmaxf(-128.f, rintf(v_new * inv_t)));
        } else {
            ((h16*)la.self_cache)[off_k] = (h16)k_new;
            ((h16*)la.self_cache)[off_v] = (h16)v_new;
        }
    }
    __syncthreads();
}

template <int WB, bool I8KV, int NR>
__global__ __launch_bounds__(512) void gemv_chain_kernel(GemvChainParams p) {
    __shared__ __attribute__((aligned(16))) float s_red[16][64][4];
    __shared__ __attribute__((aligned(16))) h16 s_in[NR > 2 ? NR : 2][CHAIN_MAX_IN + 8];
    __shared__ __attribute__((aligned(16))) h16 s_own[2 * NR][16];       // [slot][row]: the slot's 16 channels of the residual rows
    constexpr int KT = WB == 4 ? 128 : (WB == 8 ? 64 : 32);
    constexpr int TB = WB == 16 ? 10 : 5;
    // epochs never repeat: the generation word counts the decoder calls on this workspace (the embedding kernel that opens a call
    // increments it), the launch id the chains of a call, the low bits the stages of a chain.  Bit 31 is always set: the tag of a
    // zero-initialised granule (the library clears the granules of a workspace it has not seen, engine.hip) is never a valid epoch.
    const unsigned gen = CHAIN_EPOCH_LIVE | ((*p.generation & 0x1fffffu) << 10);
    extern __shared__ __attribute__((aligned(1024))) unsigned char kv_lds[];      // the cross-attention stage's K and V rows (only then)
    __shared__ float s_sc[CHAIN_CROSS_KEYS];
    __shared__ float s_redc[8][2];
    __shared__ float s_o[NR >= 4 ? 8 : 4][64];
    __shared__ float s_q[NR >= 4 ? 2 : 1][64];
    // The WHOLE token step in one launch (p.n_layers > 0): the launch walks over the layers itself.  Per-layer pointers come from two
    // tables (the engine's: biases and the cache scale; the caller's, in the workspace: cross K/V and cache of each layer), the
    // descriptors are [qkv of layer 0] + 6 per layer, and "layer -1" is that first projection alone (its sums go out as granules,
    // as every later qkv projection's do, for the self-attention stage behind it).
    const bool whole = p.n_layers > 0;
    const int T_now = p.self_t_dev ? *p.self_t_dev : p.self_T;      // cached tokens: one (scalar) load per launch
    ChainLayerArgs la{p.cross_kv, p.cross_qbias, p.self_cache, p.self_bias, p.self_kv_scale};
    const int per_split = (((p.cross_Tk + p.cross_nsplit - 1) / p.cross_nsplit) + 7) & ~7;
    // the two per-layer tables, copied to LDS once: read from memory layer by layer they were a cold scalar load (2.3 us) in front of
    // every layer's first barrier -- on the critical path in the self-attention's workgroups
    __shared__ ChainLayerStatic s_lst[CHAIN_MAX_LAYERS];
    __shared__ ChainLayerIo s_lio[CHAIN_MAX_LAYERS];
    // ... and so are the stage descriptors: a stage can request nothing before it has its descriptor, and from memory that was a
    // (vector) load of its own at the head of every stage
    constexpr int DESC_DW = (int)(sizeof(ChainStage) / 4);
    __shared__ unsigned s_desc[(1 + 6 * CHAIN_MAX_LAYERS) * DESC_DW];
    {
        const int n_desc = whole ? 6 * p.n_layers : p.n_stages;             // (whole step: [qkv of layer 0] + 6 per layer - the last layer's missing sixth)
        for (int i = threadIdx.x; i < n_desc * DESC_DW; i += 512) s_desc[i] = ((const unsigned*)p.st)[i];
        if (whole)
            for (int i = threadIdx.x; i < p.n_layers; i += 512) { s_lst[i] = p.lstat[i]; s_lio[i] = p.lio[i]; }
        __syncthreads();
    }
    // Who does what besides the Linears.  The cross-attention's (head, piece) items go to the LAST workgroups of the launch and the
    // self-attention's heads to the ones before them: the first workgroups own the output groups of every n_state-wide Linear (and
    // both slots of the widest), the last ones idle through most stages -- and the four upper waves of a workgroup that carries
    // K / V rows in flight must not meet a stage's "everything of mine has landed" wait before those rows are due.
    const int R = NR == 1 ? 1 : p.rows;
    // rows that have finished (p.live: the list of the ones still decoding): their attention stages read nothing, append nothing to their
    // cache and publish zeros; one vector load, the words handed round the wave
    unsigned dead = 0;
    if (p.live) {
        const int lv = p.live[min((int)(threadIdx.x & 63), R)];
        const int n_live = __builtin_amdgcn_readlane(lv, 0);
        unsigned alive = 0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int idx = __builtin_amdgcn_readlane(lv, 1 + i);
            if (i < n_live && i < R) alive |= 1u << (idx & 31);
        }
        dead = ~alive & ((1u << R) - 1u);
    }
    // (3 and more rows: the cross-attention's rows go from memory to registers inside its stage, two items per workgroup -- chain_cross_stage4)
    if constexpr (NR <= 2) chain_cross_prefetch<NR>(p, whole ? s_lio[0].cross_kv : p.cross_kv, kv_lds, per_split, dead);
    const int cross_items = R * p.cross_heads * p.cross_nsplit, cross_rounds = (cross_items + 511) >> 9;
    const int n_cross_wgs = NR >= 4 ? (cross_items + 2 * cross_rounds - 1) / (2 * cross_rounds) : NR * p.cross_heads * p.cross_nsplit;      // two items each (and round) | one
    const int n_self_wgs = R * p.self_heads;                                                                                   // one (row, head) each
    const int kv_avail = NR >= 4 ? (int)(NR == 4 ? CHAIN_DYN_LDS4 : CHAIN_DYN_LDS8) - 1024 : 2 * per_split * 128;      // dynamic LDS the self-attention's cached rows may take
    const int self_base = max((int)gridDim.x - n_cross_wgs - n_self_wgs, 0);
    const int self_idx = (int)blockIdx.x - self_base;                    // this workgroup's self-attention (row, head), if 0 <= self_idx < rows x heads
    // (NR == 8: a workgroup may carry a self-attention head AND cross-attention items -- different stages; nothing of the latter lives in the dynamic LDS)
    const bool self_wg = self_idx >= 0 && self_idx < n_self_wgs && (NR == 8 || (int)blockIdx.x < (int)gridDim.x - n_cross_wgs);
    const int self_h = self_wg ? self_idx % p.self_heads : 0, self_r = self_wg ? self_idx / p.self_heads : 0;
    const size_t self_row_off = (size_t)self_r * p.self_row_bytes;       // this row's share of a layer's cache
    int self_v_off = 0;                                                  // > 0: the head's cached rows of the NEXT self-attention stage are (on their way) in LDS
    const bool self_dead = self_wg && ((dead >> self_r) & 1);            // the head's row has finished: no cache append, no attention, zeros published
    if (self_wg && !self_dead) self_v_off = chain_self_prefetch<I8KV>(p, (const unsigned char*)(whole ? s_lio[0].cache : p.self_cache) + self_row_off, T_now, self_h, kv_lds, kv_avail);
    bool own_valid = false, x_in_granules = false;
    for (int l = whole ? -1 : 0; l < (whole ? p.n_layers : 1); ++l) {
        const unsigned epoch0 = gen | ((unsigned)(whole ? (l & 63) : p.launch_id) << 3);
        if (whole && l >= 0) {
            const ChainLayerStatic ls = s_lst[l];
            const ChainLayerIo li = s_lio[l];
            la.cross_kv = (const h16*)li.cross_kv; la.cross_qbias = ls.cq_bias;
            la.self_cache = li.cache; la.self_bias = ls.qkv_bias; la.self_kv_scale = ls.kv_scale;
        }
        if (l >= 0 && self_dead) {                       // (workgroup-uniform) the head's 64 outputs as zeros, tagged like the stage's own
            const int lane_ = threadIdx.x;
            // ... behind the wait the stage itself begins with (the row's q sums of the qkv stage in front): published at once, the zeros of
            // layer l could replace layer l - 1's outputs before the out projection has read them
            bool go = true;
            if (p.gran_s && lane_ < 64) {
                const unsigned tag_s = (gen | ((unsigned)((l - 1) & 63) << 3)) + 6;
                int fst[1] = {self_r * 3 * p.self_heads * 64 + self_h * 64 + 2 * min(lane_, 31)};
                u32x4 val[1];
                go = sweep_granules16<1>(p.gran_s, fst, tag_s, val, p.err, lane_);
            }
            if (go && lane_ < 64 && (lane_ & 1) == 0)
                __hip_atomic_store((chain_gu64*)(p.gran_c + ((self_r * p.self_heads * 64 + self_h * 64 + lane_) >> 1)), (unsigned long long)epoch0 << 32,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (l >= 0 && self_wg) {                  // (workgroup-uniform) LDS: the Linears' buffers, not in use now
            float* s_p = &s_red[0][0][0];
            h16 (*s_new)[64] = (h16 (*)[64])(&s_red[8][0][0]);
            float (*s_r2)[4] = (float (*)[4])(&s_red[12][0][0]);
            float* s_o_flat = (float*)&s_in[0][0];
            float* s_lut = &s_red[4][0][0];                 // 2 x 256 floats + the new key's 64 (s_red[4] .. s_red[6])
            const unsigned tag_s = (gen | ((unsigned)((l - 1) & 63) << 3)) + 6;     // the qkv stage (s = 5) of the layer before
            chain_self_stage<I8KV>(p, la, T_now, epoch0, tag_s, self_h, self_r, s_p, s_new, s_r2, s_o_flat, s_lut, kv_lds, self_v_off);
            // the NEXT layer's cached rows set out now (the stage's last barrier is behind every read of this layer's)
            self_v_off = (whole && l + 1 < p.n_layers) ? chain_self_prefetch<I8KV>(p, (const unsigned char*)s_lio[l + 1].cache + self_row_off, T_now, self_h, kv_lds, kv_avail) : 0;
        }
        const int s_first = l < 0 ? 5 : 0;
        const int s_end = !whole ? p.n_stages : (l < 0 ? 6 : (l + 1 < p.n_layers ? 6 : 5));
        for (int s = s_first; s < s_end; ++s) {
            ChainStage st;                                                     // (uniform: the descriptor's words from LDS, made scalar)
            {
                unsigned raw[DESC_DW];
                const int idx = whole ? 1 + 6 * l + s : s;
#pragma unroll
                for (int k = 0; k < DESC_DW; ++k) raw[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)s_desc[idx * DESC_DW + k]);
                __builtin_memcpy(&st, raw, sizeof(st));
            }
            const bool wide = (st.K / KT + TB - 1) / TB > 4;
            const unsigned epoch = epoch0 + (unsigned)s + 1;      // the tag this stage's results carry; its inputs carry epoch - 1
            int in_kind = CHAIN_IN_GRANULES;
            const unsigned long long* gran = p.gran_h;
            unsigned tag = epoch - 1;
            if (s == p.merge_at) {                                // (the workgroups that own a group of this stage merge the pieces inside it)
                in_kind = CHAIN_IN_LDS; tag = epoch0 + (unsigned)p.cross_at + 1;
            }
            else if (s == 0) { gran = p.gran_c; tag = epoch0; }
            if (wide) chain_stage<WB, true, false, NR>(p, st, s, epoch, own_valid, s_red, s_in, s_own, in_kind, gran, tag, x_in_granules);
            else if (st.ln_g) chain_stage<WB, false, true, NR>(p, st, s, epoch, own_valid, s_red, s_in, s_own, in_kind, gran, tag, x_in_granules);
            else chain_stage<WB, false, false, NR>(p, st, s, epoch, own_valid, s_red, s_in, s_own, in_kind, gran, tag, x_in_granules);
            if (st.mode == 2) x_in_granules = true;               // the residual row of the stages behind: this launch's granules
            if (l >= 0 && s == p.cross_at) {
                if constexpr (NR >= 4) {
                    chain_cross_stage4(p, la, epoch, per_split, s_sc, s_redc, s_o, s_q, dead);
                } else {
                    chain_cross_stage<NR>(p, la, epoch, kv_lds, per_split, s_sc, s_redc, s_o, s_q[0], dead);
                    // the NEXT layer's K / V rows set out now: they have the rest of this layer to arrive
                    if (whole && l + 1 < p.n_layers) chain_cross_prefetch<NR>(p, s_lio[l + 1].cross_kv, kv_lds, per_split, dead);
                }
            }
        }
    }
}

// the dynamic-LDS limit is a per-device attribute of each instantiation: raised once per device
static int chain_set_lds_attribute() {
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    WM_CHECK_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (attr_set.load(std::memory_order_acquire) & bit) return 0;
    auto each_kernel = [&](auto&& f) -> int {
        if (int rc = f(gemv_chain_kernel<4, false, 1>)) return rc;
        if (int rc = f(gemv_chain_kernel<4, true, 1>)) return rc;
        if (int rc = f(gemv_chain_kernel<8, false, 1>)) return rc;
        if (int rc = f(gemv_chain_kernel<8, true, 1>)) return rc;
        if (int rc = f(gemv_chain_kernel<16, false, 1>)) return rc;
        if (int rc = f(gemv_chain_kernel<16, true, 1>)) return rc;
        if (int rc = f(gemv_chain_kernel<4, false, 2>)) return rc;
        if (int rc = f(gemv_chain_kernel<4, true, 2>)) return rc;
        if (int rc = f(gemv_chain_kernel<8, false, 2>)) return rc;
        if (int rc = f(gemv_chain_kernel<8, true, 2>)) return rc;
        if (int rc = f(gemv_chain_kernel<16, false, 2>)) return rc;
        return f(gemv_chain_kernel<16, true, 2>);
    };
    if (each_kernel([&](auto* k) -> int { WM_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_DYN_LDS)); return 0; })) return 2;
    auto each_kernel4 = [&](auto&& f) -> int {       // 3 and 4 rows (8- and 16-bit weights)
        if (int rc = f(gemv_chain_kernel<8, false, 4>)) return rc;
        if (int rc = f(gemv_chain_kernel<8, true, 4>)) return rc;
        if (int rc = f(gemv_chain_kernel<16, false, 4>)) return rc;
        return f(gemv_chain_kernel<16, true, 4>);
    };
    if (each_kernel4([&](auto* k) -> int { WM_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_DYN_LDS4)); return 0; })) return 2;
    auto each_kernel8 = [&](auto&& f) -> int {       // 5 to 8 rows
        if (int rc = f(gemv_chain_kernel<8, false, 8>)) return rc;
        if (int rc = f(gemv_chain_kernel<8, true, 8>)) return rc;
        if (int rc = f(gemv_chain_kernel<16, false, 8>)) return rc;
        return f(gemv_chain_kernel<16, true, 8>);
    };
    if (each_kernel8([&](auto* k) -> int { WM_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_DYN_LDS8)); return 0; })) return 2;
    attr_set.fetch_or(bit, std::memory_order_release);
    return 0;
}

// Can the CURRENT device hold the launch's n_wg workgroups TOGETHER?  They wait for each other, so one that is never scheduled makes the
// others spin until their bounded waits give up: the runtime's occupancy figure for this instantiation at its LDS footprint (static +
// the K / V pieces) times the CUs must cover the grid.  *ok = false is an answer, not an error.
// every instantiation by its run-time selectors
template <typename F>
static auto chain_pick(int w8, bool i8, int rows, F&& f) {
    if (rows > 4) {
        if (w8) return i8 ? f(gemv_chain_kernel<8, true, 8>) : f(gemv_chain_kernel<8, false, 8>);
        return i8 ? f(gemv_chain_kernel<16, true, 8>) : f(gemv_chain_kernel<16, false, 8>);
    }
    if (rows > 2) {                                  // (4-bit weights: one or two rows only -- the callers check)
        if (w8) return i8 ? f(gemv_chain_kernel<8, true, 4>) : f(gemv_chain_kernel<8, false, 4>);
        return i8 ? f(gemv_chain_kernel<16, true, 4>) : f(gemv_chain_kernel<16, false, 4>);
    }
    if (rows == 2) {
        if (w8 == 4) return i8 ? f(gemv_chain_kernel<4, true, 2>) : f(gemv_chain_kernel<4, false, 2>);
        if (w8) return i8 ? f(gemv_chain_kernel<8, true, 2>) : f(gemv_chain_kernel<8, false, 2>);
        return i8 ? f(gemv_chain_kernel<16, true, 2>) : f(gemv_chain_kernel<16, false, 2>);
    }
    if (w8 == 4) return i8 ? f(gemv_chain_kernel<4, true, 1>) : f(gemv_chain_kernel<4, false, 1>);
    if (w8) return i8 ? f(gemv_chain_kernel<8, true, 1>) : f(gemv_chain_kernel<8, false, 1>);
    return i8 ? f(gemv_chain_kernel<16, true, 1>) : f(gemv_chain_kernel<16, false, 1>);
}

int gemv_chain_resident(int w8, int self_i8, int rows, int cross_Tk, int cross_nsplit, int n_wg, int n_cu, bool* ok, char* why, size_t why_cap) {
    *ok = false;
    WM_REQUIRE(cross_nsplit >= 1 && cross_Tk >= 1 && n_wg >= 1 && n_cu >= 1, "gemv_chain_resident: bad arguments");
    WM_REQUIRE(rows >= 1 && rows <= CHAIN_MAX_ROWS && !(rows > 2 && w8 == 4), "gemv_chain_resident: rows=%d (w8=%d)", rows, w8);
    const int per_split = (((cross_Tk + cross_nsplit - 1) / cross_nsplit) + 7) & ~7;
    const size_t dyn = rows > 4 ? CHAIN_DYN_LDS8 : (rows > 2 ? CHAIN_DYN_LDS4 : (size_t)2 * per_split * 128 + 1024);
    if (rows > 2 && per_split > 384) { if (why) snprintf(why, why_cap, "cross-attention pieces of %d keys exceed the 3- and 4-row stage's 384", per_split); return 0; }
    if (dyn > CHAIN_DYN_LDS) { if (why) snprintf(why, why_cap, "cross-attention pieces of %d keys do not fit LDS", per_split); return 0; }
    if (chain_set_lds_attribute()) return 2;
    // What one workgroup takes of a CU, from the function's own attributes, against what a CU has: 512 threads = 8 waves = 2 per SIMD
    // (512 registers per SIMD lane), static + dynamic LDS of the CU's.  The runtime's occupancy call is asked too and reported, but not
    // trusted alone: with this launch's footprint (158 of 160 KB) ROCm 7.2's answers 0 workgroups per CU for a kernel that ROCm 7.0's
    // -- the runtime torch's wheel bundles, whichever is loaded first serves the process -- and the hardware hold (profiles/r5g_*).
    hipFuncAttributes fa{};
    int per_cu_api = -1, dev = 0, lds_cu = 0, lds_blk = 0;
    const hipError_t rc = chain_pick(w8, self_i8 != 0, rows, [&](auto* k) {
        hipError_t e = hipFuncGetAttributes(&fa, (const void*)k);
        if (e == hipSuccess && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_api, (const void*)k, 512, dyn) != hipSuccess) { (void)hipGetLastError(); per_cu_api = -1; }
        return e;
    });
    WM_CHECK_HIP(rc);
    WM_CHECK_HIP(hipGetDevice(&dev));
    WM_CHECK_HIP(hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev));
    WM_CHECK_HIP(hipDeviceGetAttribute(&lds_blk, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    const size_t lds_have = (size_t)(lds_cu > lds_blk ? lds_cu : lds_blk);
    const bool lds_ok = fa.sharedSizeBytes + dyn <= lds_have, regs_ok = fa.numRegs <= 256, grid_ok = n_wg <= n_cu;
    *ok = lds_ok && regs_ok && grid_ok;
    if (why) snprintf(why, why_cap, "%zu + %zu B of LDS per workgroup (a CU has %zu), %d registers, %d workgroups on %d CUs; the runtime's occupancy call says %d per CU%s",
                      (size_t)fa.sharedSizeBytes, dyn, lds_have, fa.numRegs, n_wg, n_cu, per_cu_api, *ok ? "" : ": the launch's workgroups cannot be resident together");
    return 0;
}

bool gemv_chain_supports(int C, int w8, int n_cu) {
    // LayerNorm rows in three 16-byte pieces per lane (K <= 1536), two slots per workgroup for the widest n_state-deep stage
    // (mlp1: 4 C / 16 groups), at most 16 K slices for mlp2, one workgroup per CU
    const int KT = w8 == 4 ? 128 : (w8 ? 64 : 32), TB = w8 ? 5 : 10;
    const int slices_c = (C / KT + TB - 1) / TB, slices_4c = (4 * C / KT + TB - 1) / TB;
    return C % KT == 0 && C % 16 == 0 && C <= 1536 && slices_c <= 4 && slices_4c <= 16 && (4 * C / 16 + 1) / 2 <= n_cu && C / 16 <= n_cu;
}

// the caller's per-layer table: 8 entries per launch, by value (a larger argument block is staged through a blit under graph replay)
struct ChainIoChunk { ChainLayerIo e[8]; };
__global__ void chain_io_table_kernel(ChainLayerIo* dst, ChainIoChunk c, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = c.e[threadIdx.x];
}
int launch_chain_io_table(ChainLayerIo* dst, const ChainLayerIo* host, int n, hipStream_t stream) {
    for (int i = 0; i < n; i += 8) {
        ChainIoChunk c{};
        const int m = n - i < 8 ? n - i : 8;
        for (int k = 0; k < m; ++k) c.e[k] = host[i + k];
        hipLaunchKernelGGL(chain_io_table_kernel, dim3(1), dim3(64), 0, stream, dst + i, c, m);
    }
    WM_LAUNCH_CHECK(stream, "chain_io_table");
    return 0;
}

int launch_gemv_chain(const GemvChainParams& p, const ChainStage* hs_all, int n_wg, hipStream_t stream) {
    const bool whole = p.n_layers > 0;
    const ChainStage* hs = whole ? hs_all + 1 : hs_all;       // (whole step: the checks look at layer 0's stages; every layer has the same shapes)
    const int n_stages = whole ? (p.n_layers > 1 ? 6 : 5) : p.n_stages;
    WM_REQUIRE(n_stages >= 5 && n_stages <= CHAIN_MAX_STAGES, "gemv_chain: %d stages (a layer is out, cq, cout, mlp1, mlp2 [, qkv of the next])", n_stages);
    WM_REQUIRE(p.x && p.out32 && p.gran_x && p.gran_h && p.gran_q && p.gran_p && p.gran_c && p.err && p.generation && p.st && hs_all, "gemv_chain: null argument");
    WM_REQUIRE(p.launch_id >= 0 && p.launch_id < 128, "gemv_chain: launch_id=%d", p.launch_id);      // (7 bits under the stage index)
    WM_REQUIRE(p.cross_at == 1 && p.merge_at == 2, "gemv_chain: the cross-attention runs behind stage 1 (cq), stage 2 (cout) merges its pieces");
    WM_REQUIRE(whole ? (p.n_layers <= CHAIN_MAX_LAYERS && p.lstat && p.lio && p.gran_s && hs_all[0].ln_g && hs_all[0].mode == 0)
                     : (p.self_part && p.self_cache && p.cross_kv && (!p.self_i8 || p.self_kv_scale > 0.f)),
               "gemv_chain: %s launch: bad arguments", whole ? "whole-step" : "one-layer");
    WM_REQUIRE(p.w8 == 0 || p.w8 == 1 || p.w8 == 4, "gemv_chain: w8=%d", p.w8);
    const int KT = p.w8 == 4 ? 128 : (p.w8 ? 64 : 32);
    int widest = 0;
    for (int s = 0; s < n_stages; ++s) {
        const ChainStage& st = hs[s];
        WM_REQUIRE(st.Wt && st.K % KT == 0 && st.K <= CHAIN_MAX_IN && st.n_blocks >= 1, "gemv_chain: stage %d shape", s);
        WM_REQUIRE(st.mode >= 0 && st.mode <= 2, "gemv_chain: stage %d mode %d", s, st.mode);
        WM_REQUIRE(!st.ln_g || (st.ln_b && st.K <= 1536), "gemv_chain: stage %d LayerNorm needs beta and K <= 1536", s);
        const bool merged_in = s == p.merge_at;
        WM_REQUIRE(!merged_in || (p.merge_heads * 64 == st.K && !st.ln_g && st.n_blocks <= n_wg), "gemv_chain: merged input: %d heads for K=%d", p.merge_heads, st.K);
        WM_REQUIRE(s == 0 || st.ln_g || merged_in || hs[s - 1].mode == 1, "gemv_chain: stage %d reads the hidden row, stage %d must produce it", s, s - 1);
        WM_REQUIRE(s == 0 || !st.ln_g || hs[s - 1].mode == 2, "gemv_chain: stage %d normalises the residual row, stage %d must produce it", s, s - 1);
        const int TB = p.w8 ? 5 : 10, slices = (st.K / KT + TB - 1) / TB;
        const int need = slices > 4 ? st.n_blocks : (st.n_blocks + 1) / 2;
        widest = widest > need ? widest : need;
    }
    WM_REQUIRE(p.merge_nsplit == 4 && p.cross_nsplit == 4 && p.merge_heads == p.cross_heads, "gemv_chain: the attention pieces travel as [head][66][4]: 4 pieces");
    WM_REQUIRE(p.self_heads >= 1 && p.self_heads * 64 == hs[0].K && !hs[0].ln_g && (p.self_t_dev || (p.self_T >= 0 && p.self_T < p.self_cap)) && p.self_cap <= 512,
               "gemv_chain: self-attention stage: bad arguments");
    WM_REQUIRE(p.rows >= 1 && p.rows <= CHAIN_MAX_ROWS && !(p.rows > 2 && p.w8 == 4), "gemv_chain: rows=%d (one to four activation rows; 4-bit weights: one or two)", p.rows);
    WM_REQUIRE(p.rows == 1 || (p.cross_row_bytes > 0 && p.self_row_bytes > 0), "gemv_chain: several rows need the strides between their cross K/V and caches");
    const int cross_items = p.rows * p.cross_heads * p.cross_nsplit, cross_rounds = (cross_items + 511) >> 9;
    const int cross_wgs = p.rows > 2 ? (cross_items + 2 * cross_rounds - 1) / (2 * cross_rounds) : cross_items;      // (3 rows and more: two items per workgroup and round)
    WM_REQUIRE(p.rows > 4 ? (n_wg >= p.rows * p.self_heads && n_wg >= cross_wgs) : n_wg >= p.rows * p.self_heads + cross_wgs,
               "gemv_chain: %d workgroups for the attention stages of %d rows", n_wg, p.rows);
    WM_REQUIRE(hs[p.cross_at].mode == 0 && p.cross_Tk >= 1 && p.cross_heads * 64 == hs[p.cross_at].n_blocks * 16, "gemv_chain: cross-attention stage: bad arguments");
    const int per_split = (((p.cross_Tk + p.cross_nsplit - 1) / p.cross_nsplit) + 7) & ~7;
    WM_REQUIRE(per_split <= CHAIN_CROSS_KEYS, "gemv_chain: %d keys per piece", per_split);
    const size_t dyn = p.rows > 4 ? CHAIN_DYN_LDS8 : (p.rows > 2 ? CHAIN_DYN_LDS4 : (size_t)2 * per_split * 128 + 1024);
    WM_REQUIRE(dyn <= CHAIN_DYN_LDS && (p.rows <= 2 || per_split <= 384), "gemv_chain: cross-attention pieces of %d keys do not fit", per_split);
    widest = widest > cross_wgs ? widest : cross_wgs;
    WM_REQUIRE(n_wg >= widest, "gemv_chain: %d workgroups for stages that need %d", n_wg, widest);
    if (chain_set_lds_attribute()) return 2;
    chain_pick(p.w8, p.self_i8 != 0, p.rows, [&](auto* k) { hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), dyn, stream, p); return 0; });
    WM_LAUNCH_CHECK(stream, "gemv_chain");
    return 0;
}

// Diagnostic: n_wg workgroups that each hold `lds_bytes` of LDS and sleep for `usec` microseconds (wm_debug_occupy: the GPU test of the
// chain's give-up path keeps half of the CUs' LDS busy with it, so that a chain launch cannot get its workgroups resident together).
__global__ __launch_bounds__(64) void occupy_kernel(long long ticks, int* sink) {
    extern __shared__ unsigned char hold[];
    hold[threadIdx.x] = (unsigned char)threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (sink && hold[(threadIdx.x + 1) & 63] == 255 && ticks < 0) *sink = 1;
}
int launch_occupy(int n_wg, size_t lds_bytes, long long usec, hipStream_t stream) {
    WM_REQUIRE(n_wg >= 1 && n_wg <= 4096 && lds_bytes <= 160 * 1024 && usec >= 0 && usec <= 20000000, "wm_debug_occupy: bad arguments");
    WM_CHECK_HIP(hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(64), lds_bytes, stream, usec * 100, (int*)nullptr);       // wall_clock64: 100 MHz
    WM_LAUNCH_CHECK(stream, "occupy");
    return 0;
}

}  // namespace wm
