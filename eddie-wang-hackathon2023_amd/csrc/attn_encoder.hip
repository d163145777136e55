// Bidirectional self-attention of the Whisper audio encoder (T = 1500 frames, head size 64):
// flash-style, never materialising the T x T score matrix.
//
// Replaces the unfused QK^T / softmax / PV TensorRT ops of Attention.forward
// (R/tensorrt_llm/layers/attention.py:385-406) and W/torch_model.py:88-103.
// Arithmetic contract: q, k arrive already multiplied by d^-0.25 and rounded to fp16 (done in the
// QKV GEMM epilogue, exactly torch_model.py:93-95); scores = fp32 MFMA accumulation rounded to
// fp16; softmax in fp32 (online form); probabilities rounded to fp16; P.V fp32, rounded to fp16.
//
// gfx950 design (wave64, MFMA 16x16x32 f16):
//  * the kernel computes S^T = K . Q^T instead of Q . K^T.  In the MFMA C/D layout a lane then
//    owns ONE query (column lane & 15) and holds 4 keys per 16-key block in its registers, so
//      - the row softmax is in-register plus two cross-lane steps (xor 16, xor 32),
//      - exp(S^T) converted to fp16 IS the B operand of the second product O^T = V^T . P^T
//        (B[k = key][n = query], 8 keys per lane) -- no LDS round trip, no lane movement,
//      - the O^T accumulators share the lane -> query map, so the online rescale is per lane.
//  * V^T as the A operand comes from the row-major [key][dim] LDS tile through
//    ds_read_b64_tr_b16 (hardware transposed read): lane i of a 16-lane group receives 4 keys of
//    dim i.  The MFMA's k index is mapped to keys as k = 8g + j <-> key 16 (j >> 2) + 4g + (j & 3)
//    inside each 32-key chunk, which is exactly how the S^T accumulators are laid out.
//  * K/V tiles of 64 keys go global -> LDS directly (global_load_lds, 16 B per lane, one tile ahead, double-buffered; one
//    barrier per tile).  Rows are 128 B; the 16-byte chunk c of key row r sits at position c ^ ((r >> 1) & 7), applied
//    to the per-lane SOURCE address (the DMA writes LDS linearly) and to every read.  Round 1 staged the tiles through
//    registers into padded rows: with 246 VGPRs the 16 staging registers were spilled -- scratch stores right behind the
//    global loads, i.e. every tile waited for its successor's HBM round trip (4 scratch stores + 2 loads per tile, found in
//    the ISA in round 2).  Each wave owns QB blocks of 16 queries and re-uses every K / V fragment for all of them.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace wm {

constexpr int KT_KEYS = 64;
constexpr int AROW = 128;                      // LDS bytes per key row (64 halves, chunk-swizzled)
constexpr int KV_TILE = KT_KEYS * AROW;        // 8192

// PERSIST: fewer workgroups than (query tile, head, clip) items, each walking over its share -- the form used when the
// encoder runs on a budget of CUs beside another stream's work (wm_encoder_forward_shared): two workgroups fill a CU's
// register file, so 2 x budget of them occupy `budget` CUs and nothing else is dispatched there.  Same arithmetic.
template <int QB, bool PERSIST = false>
__global__ __launch_bounds__(256, 2) void attn_encoder_kernel(AttnEncParams p) {     // 2 waves per SIMD: <= 256 registers
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * KV_TILE];
    auto sK = [&](int b) { return smem + b * 2 * KV_TILE; };
    auto sV = [&](int b) { return smem + b * 2 * KV_TILE + KV_TILE; };

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int C = p.H * 64;
    // Items (query tile, head, clip) are numbered so that the query tiles of one (clip, head) share blockIdx % 8 -- one XCD under
    // round-robin placement (speed only) -- and follow each other closely: the head's K and V (384 KB at T = 1500) then leave
    // HBM once and serve the other query tiles from that XCD's L2 (with the tiles of a head dealt over the XCDs the kernel
    // fetched 2.2x its algorithmic bytes: profiles/r3g_pmc_stage_b192_pass2.txt).  item = 8 * j + x: head (j / nx) * 8 + x, tile j % nx.
    const int nx = (p.T + 64 * QB - 1) / (64 * QB), n_heads = p.H * p.B;
    const int n_items = 8 * ((n_heads + 7) / 8) * nx;
    int item = blockIdx.x;
    do {
    const int hh = ((item >> 3) / nx) * 8 + (item & 7), bx = (item >> 3) % nx;
    if (hh < n_heads) {                        // (wave-uniform; the padding items of the last group of eight heads)
    const int h = hh % p.H, b = hh / p.H;
    const int q_base = bx * (64 * QB) + wid * (16 * QB);
    const h16* base = p.qkv + (size_t)b * p.T * p.ld;

    // ---- Q^T fragments: B operand, lane -> query li, dims 32s + 8g .. +8 --------------------------
    half8v qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qrow = q_base + qb * 16 + li;
        if (qrow > p.T - 1) qrow = p.T - 1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qf[qb][s] = *(const half8v*)(base + (size_t)qrow * p.ld + h * 64 + s * 32 + g * 8);
    }

    // ---- K/V staging: a tile is 8 wave-wide DMA pieces (8 key rows x 128 B each) per matrix; wave w issues pieces w and
    // w + 4 of K and of V.  Lane -> key row (lane >> 3) of the piece, LDS chunk (lane & 7) <- source chunk (lane & 7) ^ swz(row)
    auto load_kv = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid + 4 * i;
            const int r = piece * 8 + (lane >> 3);                        // key row inside the tile
            int key = tile * KT_KEYS + r;
            if (key > p.T - 1) key = p.T - 1;
            const h16* row = base + (size_t)key * p.ld + h * 64 + (((lane & 7) ^ ((r >> 1) & 7)) * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + C),
                                             (__attribute__((address_space(3))) void*)(sK(buf) + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + 2 * C),
                                             (__attribute__((address_space(3))) void*)(sV(buf) + piece * 1024), 16, 0, 0);
        }
    };

    float4v o[4][QB];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[db][qb] = float4v{0.f, 0.f, 0.f, 0.f};
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m_run[qb] = -INFINITY; l_run[qb] = 0.f; }

    const int ntiles = (p.T + KT_KEYS - 1) / KT_KEYS;
    load_kv(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // LDS addresses (chunk positions XOR-swizzled with (row >> 1) & 7; a row block of 16 or 32 rows does not change that value):
    // K A-operand: row = key li (+ 16 kb), dims 8g.. (chunk g) and 32 + 8g.. (chunk g + 4)
    const int k_swz = (li >> 1) & 7;
    const int k_off0 = li * AROW + ((g ^ k_swz) << 4), k_off1 = li * AROW + (((g + 4) ^ k_swz) << 4);
    // V transposed read: lane i of its 16-lane group supplies row 4g + (i >> 2) (+ 16 for the upper half, + 32 c), 4 columns
    // from 4 (i & 3) + 16 db: chunk 2 db + ((i & 3) >> 1), its upper or lower 8 bytes
    const int v_row = 4 * g + (li >> 2), v_swz = (v_row >> 1) & 7;
    const int v_base = v_row * AROW + (li & 1) * 8, v_chunk = (li & 3) >> 1;

    auto tile_body = [&](int t, auto TAIL_) {
        constexpr bool TAIL = decltype(TAIL_)::value;     // only the last tile masks keys >= T
        const int cur = t & 1;
        if (t + 1 < ntiles) load_kv(t + 1, cur ^ 1);       // lands while this tile is multiplied; nobody reads that buffer before the barrier below

        // ---- S^T = K . Q^T : [64 keys] x [16 QB queries] ------------------------------------------
        float4v sacc[4][QB];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const half8v k0 = *(const half8v*)(sK(cur) + k_off0 + kb * 16 * AROW);
            const half8v k1 = *(const half8v*)(sK(cur) + k_off1 + kb * 16 * AROW);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float4v a = float4v{0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, qf[qb][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qf[qb][1], a, 0, 0, 0);
                sacc[kb][qb] = a;
            }
        }
        // ---- mask the tail, round to fp16, online softmax (per lane = per query) ------------------
        // The four query blocks go through every stage TOGETHER: first all local maxima, then one round of cross-lane
        // exchanges for the four of them (xor 16), then the other (xor 32) -- two LDS round trips per tile.  Taken block by
        // block (round 2) every block paid its own four dependent ds_bpermute + wait pairs: 16 exposed round trips per
        // tile, more wave time than the tile's 64 MFMAs.  The row SUMS are not exchanged per tile at all: a lane keeps the
        // partial sum over the keys it owns (the rescale factor alpha is common to a query's four lanes), the four partial
        // sums meet once, after the last tile.
        const int key0 = t * KT_KEYS + 4 * g;          // + 16 kb + r
        half8v pf[2][QB];                              // P^T fragments per 32-key chunk
        constexpr float LOG2E = 1.4426950408889634f;
        float mx[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            // the maximum is taken on the raw scores and rounded once (rounding is monotonic)
            float m = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (TAIL && key0 + 16 * kb + r >= p.T) sacc[kb][qb][r] = -INFINITY;
                    m = fmaxf(m, sacc[kb][qb][r]);
                }
            mx[qb] = m;
        }
        {
            float o16[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) o16[qb] = __shfl_xor(mx[qb], 16);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) mx[qb] = fmaxf(mx[qb], o16[qb]);
            float o32[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) o32[qb] = __shfl_xor(mx[qb], 32);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) mx[qb] = fmaxf(mx[qb], o32[qb]);
        }
        float mL[QB], alpha[QB];
        bool moved = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float m_old = m_run[qb];
            const float m_new = fmaxf(m_old, r16(mx[qb]));
            // explicit roundings (no fp-contract freedom): both template instantiations must give the same bits,
            // the result of a clip must not depend on how many clips share the launch
            // (HIP's __fmul_rn / __fsub_rn are plain operations under -ffp-contract=fast: in another context hipcc fused this multiply and
            // subtraction into one v_fma_f32 and 0.5 % of the outputs moved by an fp16 ulp -- profiles/r6v_*.  The empty asm pins both products.)
            float mn = __fmul_rn(m_new, LOG2E), mo = __fmul_rn(m_old, LOG2E);
            asm volatile("" : "+v"(mn), "+v"(mo));              // (pinning only one of them made hipcc fuse the OTHER product into the subtraction)
            mL[qb] = mn;
            alpha[qb] = __builtin_amdgcn_exp2f(__fsub_rn(mo, mL[qb]));   // exp(m_old - m_new); 0 on the first tile
            moved |= (m_new != m_old);
            m_run[qb] = m_new;
        }
        const bool any_moved = __builtin_amdgcn_ballot_w64(moved) != 0;     // wave-uniform
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            // scores and probabilities are rounded to fp16 in pairs (v_cvt_pk_f16_f32), exp(s - m) is one mixed-precision
            // fma + v_exp_f32 on the fp16 score, the row sum one v_dot2 per pair
            float ps = 0.f;
            const half2v one2 = {(h16)1.0f, (h16)1.0f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const half2v s2 = __builtin_convertvector(float2v{sacc[kb][qb][r], sacc[kb][qb][r + 1]}, half2v);
                    const float p0 = __builtin_amdgcn_exp2f(__fmaf_rn((float)s2[0], LOG2E, -mL[qb]));
                    const float p1 = __builtin_amdgcn_exp2f(__fmaf_rn((float)s2[1], LOG2E, -mL[qb]));
                    const half2v p2 = __builtin_convertvector(float2v{p0, p1}, half2v);
                    ps = __builtin_amdgcn_fdot2(p2, one2, ps, false);
                    pf[kb >> 1][qb][(kb & 1) * 4 + r] = p2[0];
                    pf[kb >> 1][qb][(kb & 1) * 4 + r + 1] = p2[1];
                }
            l_run[qb] = __fmaf_rn(l_run[qb], alpha[qb], ps);     // this lane's keys only; the four lanes of a query meet at the end
        }
        // (Round 4 tried the test per block of 16 queries -- a block's maxima move half as often as the wave's 64 -- : four ballots
        // and branches per tile instead of one, 1.5 % SLOWER at B = 128 / 256, profiles/r4a_attn_rescale_per_block_ab.log.)
        if (any_moved) {                            // skip the accumulator round trip while no query's maximum moved (alpha == 1 exactly)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db][qb] *= alpha[qb];
        }
        // ---- O^T += V^T . P^T ------------------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const unsigned char* va = sV(cur) + v_base + (c * 32) * AROW + (((2 * db + v_chunk) ^ v_swz) << 4);
                const short4r lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4r*)(va));
                const short4r hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4r*)(va + 16 * AROW));
                half8v vf;
                const half4v l4 = __builtin_bit_cast(half4v, lo), h4 = __builtin_bit_cast(half4v, hi);
#pragma unroll
                for (int j = 0; j < 4; ++j) { vf[j] = l4[j]; vf[4 + j] = h4[j]; }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[c][qb], o[db][qb], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next tile have landed
        __syncthreads();
    };
    for (int t = 0; t + 1 < ntiles; ++t) tile_body(t, std::false_type{});
    if (p.T % KT_KEYS != 0) tile_body(ntiles - 1, std::true_type{}); else tile_body(ntiles - 1, std::false_type{});

    // ---- normalise and store: lane owns query li, dims 16 db + 4 g + r ------------------------------
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q_base + qb * 16 + li;
        float l = l_run[qb];                                     // the four lanes of a query hold the sums over their own keys
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (qrow >= p.T) continue;
        const float inv = 1.0f / l;
        h16* dst = p.out + ((size_t)b * p.T + qrow) * p.ldo + h * 64;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            half4v w;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // fp32 product, THEN fp16.  Left alone, the compiler turns some of these (which ones differs between the
                // two template instantiations) into v_fma_mixlo_f16, one rounding instead of two: the output of a clip
                // would depend on how many clips share the launch.  The empty asm pins the fp32 value.
                float t = o[db][qb][r] * inv;
                asm volatile("" : "+v"(t));
                w[r] = (h16)t;
            }
            *(half4v*)(dst + db * 16 + g * 4) = w;
        }
    }
    }
    item += gridDim.x;
    } while (PERSIST && item < n_items);
}

int launch_attn_encoder(const AttnEncParams& p, hipStream_t stream) {
    WM_REQUIRE(p.T >= 1 && p.H >= 1 && p.B >= 1, "attn_encoder: bad shape");
    WM_REQUIRE(p.ld % 8 == 0 && p.ldo % 4 == 0, "attn_encoder: leading dimensions must keep 16-byte alignment");
    // QB = 4 (256 queries per workgroup, half the LDS traffic per MFMA) whenever the sequence fills such a tile; it is
    // the faster variant at every batch size for T = 1500 (B = 1: 60 vs 67 us, B = 128: 21 vs 36 us per clip-layer).
    // The choice depends on T only, never on the batch: the two instantiations agree to one fp16 ulp, not bit for bit,
    // and the result of a clip must not depend on how many clips share the launch (tests: batch independence).
    static const int lab_wgs = lab_env_int("WM_ATTN_MAX_WGS", 0);        // probes only (scripts/kv_beside_probe.py; honoured under WM_LAB=1)
    const int max_wgs = p.max_wgs > 0 ? p.max_wgs : lab_wgs;
    const int heads8 = 8 * ((p.H * p.B + 7) / 8);            // items: see the kernel (8 heads x nx query tiles per group)
    if (p.T > 128 && max_wgs > 0 && max_wgs < ((p.T + 255) / 256) * p.H * p.B) {
        hipLaunchKernelGGL((attn_encoder_kernel<4, true>), dim3(max_wgs >= 8 ? max_wgs / 8 * 8 : max_wgs), dim3(256), 0, stream, p);
    } else if (p.T > 128) {
        hipLaunchKernelGGL(attn_encoder_kernel<4>, dim3(heads8 * ((p.T + 255) / 256)), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL(attn_encoder_kernel<2>, dim3(heads8 * ((p.T + 127) / 128)), dim3(256), 0, stream, p);
    }
    WM_LAUNCH_CHECK(stream, "attn_encoder");
    return 0;
}

}  // namespace wm
