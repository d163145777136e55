// Decode-time attention for the Whisper decoder: self-attention over the (int8 or fp16) KV cache
// with in-place append, and cross-attention over the per-utterance fp16 cross K/V.
//
// Arithmetic contract (what the oracle restates):
//   q, k are multiplied by d^-0.25 and rounded to fp16 (W/torch_model.py:93-95), scores are
//   fp32 dot products rounded to fp16, softmax runs in fp32 over the whole key range
//   (torch_model.py:100-101; attention.py:385-398), probabilities are rounded to fp16, P.V
//   accumulates in fp32 and is rounded to fp16.  The self-attention's dot products and P.V sums are chains of fused
//   multiply-adds (fmaf, written out: left to the compiler the same loop came out partly fused, partly not, differently in
//   every copy of it -- and gemv_chain.hip runs a copy of the four-wave form that must agree with it bit for bit).
//   int8 KV: present = sat_s8(rne(x * (1/t))), past is used as fp16(q8) * t, the tokens of the
//   current call are used un-quantised (attention.py:281-348; same contract as the MMHA kernel,
//   decoderMaskedMultiheadAttentionTemplate.h:1501-1517, Utils.h:2276-2286,2357-2390).
//
// Design reference for the single-token kernel: MaskedMultiheadAttention
// (R/cpp/tensorrt_llm/kernels/decoderMaskedMultiheadAttention/...Template.h:1195-2188): one
// block per (head, sequence), q.K over the cache, block softmax, V accumulation.  Here:
//
// * self-attention: one wave per (b, h).  <= 448 keys: lane-per-key dot products (a key row is
//   64 B int8 / 128 B fp16, neighbouring lanes read neighbouring rows), wave-shuffle softmax,
//   lane-per-dim V accumulation.  The cache append happens in the same kernel.
// * cross-attention: the dominant HBM stream of a batched decode step (2 * H * 1500 * 64 * 2 B
//   = 7.68 MB per utterance per layer at large-v2).  One workgroup of 256 threads per
//   (b, h, key-split).  K and V are streamed with wave-wide 16-byte loads: one load instruction
//   covers 8 whole rows (1 KiB contiguous), 8 lanes share a row and combine their partial dot
//   products with three DPP-style shuffles.  Scores for the whole key range live in LDS, so the
//   softmax is the exact two-pass one (no online rescaling), then V is streamed the same way.
#include <hip/hip_ext.h>

#include <atomic>

#include "common.h"
#include "kernels.h"

namespace wm {

constexpr float ATTN_SCALE = 0.35355339059327373f;    // 64^-0.25
constexpr int MAX_L = 4;                              // query tokens per call handled per pass

// ------------------------------------------------------------------------------------------------
// self-attention
// ------------------------------------------------------------------------------------------------
template <bool I8>
__global__ __launch_bounds__(64) void attn_self_kernel(AttnSelfParams p) {
    // issue priority over the other groups' K/V stream waves (see gemm_skinny_kernel)
    __builtin_amdgcn_s_setprio(3);

    constexpr int MAXT = 512;
    __shared__ float s_p[MAXT];
    __shared__ h16 s_qall[MAX_L][64], s_knew[MAX_L][64], s_vnew[MAX_L][64];

    const int h = blockIdx.x, lane = threadIdx.x;
    int b = blockIdx.y;
    // the live-row count, this row's entry of the list and the device-resident step counter: three UNCONDITIONAL loads in flight
    // together (each behind its own test they were three round trips in a row in front of the cache prefetch); an absent list /
    // counter reads a word of the qkv sums instead and drops it
    const int32_t* lsrc = p.live ? p.live : (const int32_t*)p.part;
    const int32_t live_n = lsrc[0], live_b = lsrc[p.live ? 1 + b : 0];
    const int32_t t_now = (p.t_dev ? p.t_dev : (const int32_t*)p.part)[0];
    if (p.live) {                                // rows still decoding (wave-uniform): the others' caches are not touched
        if (b >= live_n) return;
        b = live_b;
    }
    const int T = p.t_dev ? t_now : p.T;         // device-resident step counter (graph replay) or host value
    const int C = p.H * 64;
    // The kernel is a chain of dependent memory round trips (this call's q / k / v sums, the cached K rows, the cached V rows)
    // and at small batches nothing else: the first 64 K rows (one per lane) and the first 32 V rows (lane = dim) do not depend on
    // q, so they are requested before anything is waited for -- for T <= 32 cached tokens the whole kernel is one round trip.
    const unsigned char* pastK0 = (const unsigned char*)p.past + ((size_t)b * p.past_bstride + (size_t)(0 * p.H + h) * p.past_cap * 64) * (I8 ? 1 : 2);
    const unsigned char* pastV0 = (const unsigned char*)p.past + ((size_t)b * p.past_bstride + (size_t)(1 * p.H + h) * p.past_cap * 64) * (I8 ? 1 : 2);
    // (no per-element test around a load: hipcc would branch around each and wait for it before the next; rows past the
    // end re-read the last cached row instead, their values are never used)
    uint4 kpre[I8 ? 4 : 8];
    int8_t vpre8[32]; h16 vpre16[32];
    if (T > 0) {                                 // wave-uniform
        const int kr = min(lane, T - 1);
#pragma unroll
        for (int c = 0; c < (I8 ? 4 : 8); ++c) kpre[c] = ((const uint4*)(pastK0 + (size_t)kr * (I8 ? 64 : 128)))[c];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int vr = min(u, T - 1);
            if (I8) vpre8[u] = ((const int8_t*)pastV0)[(size_t)vr * 64 + lane];
            else vpre16[u] = ((const h16*)pastV0)[(size_t)vr * 64 + lane];
        }
    }
    const float t_dq = p.kv_scale;
    const float inv_t = 1.0f / p.kv_scale;
    const size_t sstride = p.part_sstride ? (size_t)p.part_sstride : (size_t)p.B * p.L * p.ldp;

    // ---- this call's q, k, v for the head: sum the split-K slabs, add bias, round to fp16 --------
    // lane = head dim
    for (int i = 0; i < p.L; ++i) {
        const int m = b * p.L + i;
        float q = 0.f, k = 0.f, v = 0.f;
        // (the three biases: unconditional loads in flight with the slabs -- behind `p.bias ? ... : 0` each was waited for inside its
        // branch, three round trips in a row; without a bias a few bytes of the slab are read and dropped)
        const h16* bsrc = p.bias ? p.bias + h * 64 + lane : (const h16*)p.part;
        const int bstep = p.bias ? C : 0;
        const h16 bq_raw = bsrc[0], bk_raw = bsrc[bstep], bv_raw = bsrc[2 * bstep];
        {
            const float* row = p.part + (size_t)m * p.ldp + h * 64 + lane;
            int s = 0;
            for (; s + 4 <= p.ksplit; s += 4) {        // 12 independent loads in flight (L2-latency bound)
                const float* r0 = row + (size_t)s * sstride;
                const float* r1 = r0 + sstride; const float* r2 = r1 + sstride; const float* r3 = r2 + sstride;
                const float q0 = r0[0], q1 = r1[0], q2 = r2[0], q3 = r3[0];
                const float k0 = r0[C], k1 = r1[C], k2 = r2[C], k3 = r3[C];
                const float v0 = r0[2 * C], v1 = r1[2 * C], v2 = r2[2 * C], v3 = r3[2 * C];
                q += (q0 + q1) + (q2 + q3); k += (k0 + k1) + (k2 + k3); v += (v0 + v1) + (v2 + v3);
            }
            for (; s < p.ksplit; ++s) {
                const float* r0 = row + (size_t)s * sstride;
                q += r0[0]; k += r0[C]; v += r0[2 * C];
            }
        }
        q = r16(q + (p.bias ? (float)bq_raw : 0.f));
        k = r16(k + (p.bias ? (float)bk_raw : 0.f));
        v = r16(v + (p.bias ? (float)bv_raw : 0.f));
        if (p.amax) {     // calibration hook: max |q|,|k|,|v| of this layer (smoothquant.py:117-175, F8)
            const float a = wave_max_nomfma(fmaxf(fabsf(q), fmaxf(fabsf(k), fabsf(v))));
            if (lane == 0) atomicMax((unsigned int*)p.amax, __float_as_uint(a));   // a >= 0: bit order = value order
        }
        s_knew[i][lane] = (h16)k;
        s_vnew[i][lane] = (h16)v;
        // append to the cache (present), position T + i
        const size_t off_k = (size_t)b * p.present_bstride + ((size_t)(0 * p.H + h) * p.present_cap + T + i) * 64 + lane;
        const size_t off_v = (size_t)b * p.present_bstride + ((size_t)(1 * p.H + h) * p.present_cap + T + i) * 64 + lane;
        if (I8) {
            ((int8_t*)p.present)[off_k] = (int8_t)fminf(127.f, fmaxf(-128.f, rintf(k * inv_t)));
            ((int8_t*)p.present)[off_v] = (int8_t)fminf(127.f, fmaxf(-128.f, rintf(v * inv_t)));
        } else {
            ((h16*)p.present)[off_k] = (h16)k;
            ((h16*)p.present)[off_v] = (h16)v;
        }
        // stash q rows in registers via LDS later; keep q of token i in s_q when processed
        s_qall[i][lane] = (h16)r16(q * ATTN_SCALE);
    }
    // copy-forward: when present is a different buffer than past (the reference's concat
    // semantics, attention.py:296-306), move the T cached rows of this head
    const bool inplace = (p.past == p.present) && (p.past_cap == p.present_cap) && (p.past_bstride == p.present_bstride);
    if (!inplace && T > 0) {
        const int es = I8 ? 1 : 2;
        for (int kv = 0; kv < 2; ++kv) {
            const unsigned char* src = (const unsigned char*)p.past + ((size_t)b * p.past_bstride + (size_t)(kv * p.H + h) * p.past_cap * 64) * es;
            unsigned char* dst = (unsigned char*)p.present + ((size_t)b * p.present_bstride + (size_t)(kv * p.H + h) * p.present_cap * 64) * es;
            const int n16 = T * 64 * es / 16;
            for (int c = lane; c < n16; c += 64) ((uint4*)dst)[c] = ((const uint4*)src)[c];
        }
    }
    __syncthreads();

    const unsigned char* pastK = (const unsigned char*)p.past + ((size_t)b * p.past_bstride + (size_t)(0 * p.H + h) * p.past_cap * 64) * (I8 ? 1 : 2);
    const unsigned char* pastV = (const unsigned char*)p.past + ((size_t)b * p.past_bstride + (size_t)(1 * p.H + h) * p.past_cap * 64) * (I8 ? 1 : 2);

    for (int i = 0; i < p.L; ++i) {
        const h16* s_q = s_qall[i];
        const int nk = T + i + 1;                 // causal: past + new tokens 0..i
        // ---- scores: lane-per-key -------------------------------------------------------------
        float mx = -INFINITY;
        for (int j0 = 0; j0 < nk; j0 += 64) {
            const int j = j0 + lane;
            float sc = -INFINITY;
            if (j < nk) {
                float acc = 0.f;
                if (j < T) {
                    if (I8) {
                        const uint4* kr = (const uint4*)(pastK + (size_t)j * 64);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint4 w = j0 == 0 ? kpre[c] : kr[c];
                            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int8_t q8 = (int8_t)((ws[e >> 2] >> (8 * (e & 3))) & 0xff);
                                const float kd = r16(r16((float)q8 * t_dq) * ATTN_SCALE);
                                acc = fmaf((float)s_q[c * 16 + e], kd, acc);
                            }
                        }
                    } else {
                        const half8v* kr = (const half8v*)(pastK + (size_t)j * 128);
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const half8v w = j0 == 0 ? __builtin_bit_cast(half8v, kpre[c]) : kr[c];
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                acc = fmaf((float)s_q[c * 8 + e], r16((float)w[e] * ATTN_SCALE), acc);
                        }
                    }
                } else {
                    const h16* kn = s_knew[j - T];
#pragma unroll 8
                    for (int e = 0; e < 64; ++e) acc = fmaf((float)s_q[e], r16((float)kn[e] * ATTN_SCALE), acc);
                }
                sc = r16(f32_as_is(acc));
            }
            if (j < MAXT) s_p[j] = sc;
            mx = fmaxf(mx, sc);
        }
        mx = wave_max_nomfma(mx);
        float sum = 0.f;
        __syncthreads();
        for (int j = lane; j < nk; j += 64) {
            const float e = __expf(s_p[j] - mx);
            s_p[j] = e;
            sum += e;
        }
        sum = wave_sum_nomfma(sum);
        const float inv = 1.0f / sum;
        __syncthreads();
        for (int j = lane; j < nk; j += 64) s_p[j] = r16(s_p[j] * inv);
        __syncthreads();
        // ---- P.V: lane-per-dim ----------------------------------------------------------------
        float o = 0.f;
        int j = 0;
        if (I8) {
            const int8_t* pv = (const int8_t*)pastV + lane;
#pragma unroll
            for (int u = 0; u < 32; ++u)          // the prefetched rows (same ascending order as the loops below)
                if (u < T) o = fmaf(s_p[u], r16((float)vpre8[u] * t_dq), o);
            j = min(T, 32);
            for (; j + 32 <= T; j += 32) {        // 32 loads in flight per lane: the loop is HBM-latency bound
                int8_t vq[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) vq[u] = pv[(size_t)(j + u) * 64];
#pragma unroll
                for (int u = 0; u < 32; ++u) o = fmaf(s_p[j + u], r16((float)vq[u] * t_dq), o);
            }
            for (; j + 8 <= T; j += 8) {
                int8_t vq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vq[u] = pv[(size_t)(j + u) * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) o = fmaf(s_p[j + u], r16((float)vq[u] * t_dq), o);
            }
            for (; j < T; ++j) o = fmaf(s_p[j], r16((float)pv[(size_t)j * 64] * t_dq), o);
        } else {
            const h16* pv = (const h16*)pastV + lane;
#pragma unroll
            for (int u = 0; u < 32; ++u)          // the prefetched rows (same ascending order as the loops below)
                if (u < T) o = fmaf(s_p[u], (float)vpre16[u], o);
            j = min(T, 32);
            for (; j + 32 <= T; j += 32) {
                h16 vh[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) vh[u] = pv[(size_t)(j + u) * 64];
#pragma unroll
                for (int u = 0; u < 32; ++u) o = fmaf(s_p[j + u], (float)vh[u], o);
            }
            for (; j + 8 <= T; j += 8) {
                h16 vh[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) vh[u] = pv[(size_t)(j + u) * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) o = fmaf(s_p[j + u], (float)vh[u], o);
            }
            for (; j < T; ++j) o = fmaf(s_p[j], (float)pv[(size_t)j * 64], o);
        }
        for (; j < nk; ++j) o = fmaf(s_p[j], (float)s_vnew[j - T][lane], o);
        p.out[(size_t)(b * p.L + i) * p.ldo + h * 64 + lane] = (h16)f32_as_is(o);
        __syncthreads();
    }
}

// The same attention with FOUR waves per (b, h) -- the form small groups take (launch_attn_self).  With one wave per head a
// token step at 64-128 cached tokens is three to five dependent memory round trips plus ~3 us of arithmetic in that single wave
// (64 dequantised products per key and lane, one fused multiply-add per cached V row), and at a few rows per group nothing else
// runs on the chip meanwhile (profiles/r3ab_b1_kernel_stats.csv: 8.7 us per launch at B = 1, the longest link of the small-batch
// chain).  Here the key range is dealt over four waves in blocks -- keys in blocks of 64 (a key row per lane), V rows by wave-wide
// 16-byte loads that cover whole rows (16 int8 rows or 8 fp16 rows = 1 KiB per instruction; the one-wave form reads one BYTE per
// lane and row) -- and every wave requests its first K block and its first four V blocks before anything else: up to 256 cached
// tokens (128 with an fp16 cache) the cache is read in ONE round trip that overlaps the q / k / v sums.  The waves meet in LDS:
// maximum, sum, and the partial P.V sums (added in a fixed order: wave, row inside a block, block).  Same rounding points as the
// one-wave form (scores, probabilities and the dequantised cache values are rounded to fp16, sums are fp32); the fp32 additions
// of the softmax sum and of P.V happen in a different order, so the two forms agree to fp32 rounding, not bit for bit.
constexpr int SELF_WAVES = 4;
template <bool I8>
__global__ __launch_bounds__(64 * SELF_WAVES) void attn_self_wg_kernel(AttnSelfParams p) {
    __builtin_amdgcn_s_setprio(3);
    constexpr int MAXT = 512;
    constexpr int ES = I8 ? 1 : 2;                // bytes per cache element
    constexpr int ROW_B = 64 * ES;                // bytes per cached row
    constexpr int DIMS = 16 / ES;                 // head dims per 16-byte chunk: 16 | 8
    constexpr int NCH = 64 / DIMS;                // chunks per row: 4 | 8
    constexpr int VROWS = 64 / NCH;               // rows per wave-wide V load: 16 | 8
    constexpr int KCH = ROW_B / 16;               // 16-byte chunks of a key row: 4 | 8
    constexpr int VPRE = 4;                       // V loads per wave requested up front
    constexpr int NT = 64 * SELF_WAVES;
    __shared__ float s_p[MAXT];
    __shared__ h16 s_qall[MAX_L][64], s_knew[MAX_L][64], s_vnew[MAX_L][64];
    __shared__ float s_o[NT][DIMS + 1];
    __shared__ float s_red[2][SELF_WAVES];
    // int8 cache: what a cached code contributes depends on the code and the layer's scale only (r16(r16(code t) SCALE) to a score,
    // r16(code t) to P.V): 2 x 256 values, computed once per launch and looked up -- the same bits as evaluating the expression per
    // element (seven vector instructions each, ~ a microsecond of a one-wave-per-64-keys score pass; csrc/gemv_chain.hip does the same)
    __shared__ float s_lut[I8 ? 512 + MAX_L * 64 : 1];      // ... and behind them the 64 score factors r16(k * SCALE) of each NEW key

    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int b = blockIdx.y;
    const int32_t* lsrc = p.live ? p.live : (const int32_t*)p.part;      // (three unconditional loads in flight together: see attn_self_kernel)
    const int32_t live_n = lsrc[0], live_b = lsrc[p.live ? 1 + b : 0];
    const int32_t t_now = (p.t_dev ? p.t_dev : (const int32_t*)p.part)[0];
    if (p.live) {                                 // rows still decoding (workgroup-uniform)
        if (b >= live_n) return;
        b = live_b;
    }
    const int T = p.t_dev ? t_now : p.T;
    const int C = p.H * 64;
    const unsigned char* pastK = (const unsigned char*)p.past + ((size_t)b * p.past_bstride + (size_t)(0 * p.H + h) * p.past_cap * 64) * ES;
    const unsigned char* pastV = (const unsigned char*)p.past + ((size_t)b * p.past_bstride + (size_t)(1 * p.H + h) * p.past_cap * 64) * ES;
    // rows past the end re-read the last cached row (no branch around a load); their values are never used
    const int vr = lane / NCH, vc = lane % NCH;   // V loads: row inside the block, 16-byte chunk of the row
    uint4 kpre[KCH], vpre[VPRE];
    if (T > 0) {                                  // workgroup-uniform
        const int kr = min(64 * wid + lane, T - 1);
#pragma unroll
        for (int c = 0; c < KCH; ++c) kpre[c] = ((const uint4*)(pastK + (size_t)kr * ROW_B))[c];
#pragma unroll
        for (int n = 0; n < VPRE; ++n) {
            const int row = min((wid + SELF_WAVES * n) * VROWS + vr, T - 1);
            vpre[n] = *(const uint4*)(pastV + (size_t)row * ROW_B + vc * 16);
        }
    }
    const float t_dq = p.kv_scale;
    const float inv_t = 1.0f / p.kv_scale;
    if constexpr (I8) {                           // (256 threads: one code each, both tables; the barrier below covers it)
        const float d = r16((float)(int)(int8_t)(tid & 255) * t_dq);
        s_lut[tid & 255] = r16(d * ATTN_SCALE);
        s_lut[256 + (tid & 255)] = d;
    }
    const size_t sstride = p.part_sstride ? (size_t)p.part_sstride : (size_t)p.B * p.L * p.ldp;

    // ---- this call's q, k, v for the head (wave 0; lane = head dim): slabs summed, bias, fp16; cache append -------------------
    if (wid == 0) {
        for (int i = 0; i < p.L; ++i) {
            const int m = b * p.L + i;
            float q = 0.f, k = 0.f, v = 0.f;
            const h16* bsrc = p.bias ? p.bias + h * 64 + lane : (const h16*)p.part;      // (unconditional bias loads: see attn_self_kernel)
            const int bstep = p.bias ? C : 0;
            const h16 bq_raw = bsrc[0], bk_raw = bsrc[bstep], bv_raw = bsrc[2 * bstep];
            const float* row = p.part + (size_t)m * p.ldp + h * 64 + lane;
            int s = 0;
            for (; s + 4 <= p.ksplit; s += 4) {    // the one-wave form's order of additions
                const float* r0 = row + (size_t)s * sstride;
                const float* r1 = r0 + sstride; const float* r2 = r1 + sstride; const float* r3 = r2 + sstride;
                const float q0 = r0[0], q1 = r1[0], q2 = r2[0], q3 = r3[0];
                const float k0 = r0[C], k1 = r1[C], k2 = r2[C], k3 = r3[C];
                const float v0 = r0[2 * C], v1 = r1[2 * C], v2 = r2[2 * C], v3 = r3[2 * C];
                q += (q0 + q1) + (q2 + q3); k += (k0 + k1) + (k2 + k3); v += (v0 + v1) + (v2 + v3);
            }
            for (; s < p.ksplit; ++s) {
                const float* r0 = row + (size_t)s * sstride;
                q += r0[0]; k += r0[C]; v += r0[2 * C];
            }
            q = r16(q + (p.bias ? (float)bq_raw : 0.f));
            k = r16(k + (p.bias ? (float)bk_raw : 0.f));
            v = r16(v + (p.bias ? (float)bv_raw : 0.f));
            if (p.amax) {     // calibration hook (see attn_self_kernel)
                const float a = wave_max_nomfma(fmaxf(fabsf(q), fmaxf(fabsf(k), fabsf(v))));
                if (lane == 0) atomicMax((unsigned int*)p.amax, __float_as_uint(a));
            }
            s_knew[i][lane] = (h16)k;
            s_vnew[i][lane] = (h16)v;
            if constexpr (I8) s_lut[512 + i * 64 + lane] = r16(k * ATTN_SCALE);      // (k is an fp16 value already)
            const size_t off_k = (size_t)b * p.present_bstride + ((size_t)(0 * p.H + h) * p.present_cap + T + i) * 64 + lane;
            const size_t off_v = (size_t)b * p.present_bstride + ((size_t)(1 * p.H + h) * p.present_cap + T + i) * 64 + lane;
            if (I8) {
                ((int8_t*)p.present)[off_k] = (int8_t)fminf(127.f, fmaxf(-128.f, rintf(k * inv_t)));
                ((int8_t*)p.present)[off_v] = (int8_t)fminf(127.f, fmaxf(-128.f, rintf(v * inv_t)));
            } else {
                ((h16*)p.present)[off_k] = (h16)k;
                ((h16*)p.present)[off_v] = (h16)v;
            }
            s_qall[i][lane] = (h16)r16(q * ATTN_SCALE);
        }
    }
    // copy-forward when present is a different buffer than past (the reference's concat semantics, attention.py:296-306)
    const bool inplace = (p.past == p.present) && (p.past_cap == p.present_cap) && (p.past_bstride == p.present_bstride);
    if (!inplace && T > 0) {
        for (int kv = 0; kv < 2; ++kv) {
            const unsigned char* src = (const unsigned char*)p.past + ((size_t)b * p.past_bstride + (size_t)(kv * p.H + h) * p.past_cap * 64) * ES;
            unsigned char* dst = (unsigned char*)p.present + ((size_t)b * p.present_bstride + (size_t)(kv * p.H + h) * p.present_cap * 64) * ES;
            const int n16 = T * 64 * ES / 16;
            for (int c = tid; c < n16; c += NT) ((uint4*)dst)[c] = ((const uint4*)src)[c];
        }
    }
    __syncthreads();

    for (int i = 0; i < p.L; ++i) {
        const h16* s_q = s_qall[i];
        const int nk = T + i + 1;                 // causal: past + new tokens 0..i
        // ---- scores: key blocks of 64 dealt over the waves, a key per lane ------------------------------------------------
        float mx = -INFINITY;
        for (int kb = wid; kb * 64 < nk; kb += SELF_WAVES) {
            const int j = kb * 64 + lane;
            float sc = -INFINITY;
            if (j < nk) {
                // ONE chain of 64 products per key, cached or new: a new key's lane (j >= T) takes its factors from the values wave 0 left
                // behind the tables (int8) / from the new row in LDS (fp16) by a per-lane select -- as a branch of its own its 64 products
                // were a second pass behind the cached keys' for a handful of lanes.  Same factors, same order of additions.
                float acc = 0.f;
                const bool cached = j < T;
                const int jn = cached ? 0 : j - T;
                const uint4* kr = (const uint4*)(pastK + (size_t)min(j, max(T - 1, 0)) * ROW_B);
#pragma unroll
                for (int c = 0; c < KCH; ++c) {
                    const uint4 w = (kb == wid && T > 0) ? kpre[c] : kr[c];
                    if (I8) {
                        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int idx = cached ? (int)((ws[e >> 2] >> (8 * (e & 3))) & 0xff) : 512 + jn * 64 + c * 16 + e;
                            acc = fmaf((float)s_q[c * 16 + e], s_lut[idx], acc);      // r16(r16((float)code * t_dq) * ATTN_SCALE) | r16(k_new * ATTN_SCALE)
                        }
                    } else {
                        const half8v wh = __builtin_bit_cast(half8v, w);
                        const half8v kn8 = *(const half8v*)(&s_knew[jn][c * 8]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xk = cached ? (float)wh[e] : (float)kn8[e];
                            acc = fmaf((float)s_q[c * 8 + e], r16(xk * ATTN_SCALE), acc);
                        }
                    }
                }
                sc = r16(f32_as_is(acc));
                s_p[j] = sc;
            }
            mx = fmaxf(mx, sc);
        }
        mx = wave_max_nomfma(mx);
        if (lane == 0) s_red[0][wid] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
        // ---- softmax: thread t owns keys t and t + 256 ---------------------------------------------------------------------
        const float e0 = tid < nk ? __expf(s_p[tid] - mx) : 0.f;
        const float e1 = tid + NT < nk ? __expf(s_p[tid + NT] - mx) : 0.f;
        const float wsum = wave_sum_nomfma(e0 + e1);
        if (lane == 0) s_red[1][wid] = wsum;
        __syncthreads();
        const float inv = 1.0f / ((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]));
        if (tid < nk) s_p[tid] = r16(e0 * inv);
        if (tid + NT < nk) s_p[tid + NT] = r16(e1 * inv);
        __syncthreads();
        // ---- P.V over the cached rows: blocks of VROWS rows dealt over the waves; a lane holds DIMS dims of one row per block --
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
            const int vb = wid + SELF_WAVES * n;
            if (vb * VROWS < T) add_block(vb, vpre[n]);           // (wave-uniform)
        }
        for (int vb = wid + SELF_WAVES * VPRE; vb * VROWS < T; vb += SELF_WAVES) {
            const int row = min(vb * VROWS + vr, T - 1);
            add_block(vb, *(const uint4*)(pastV + (size_t)row * ROW_B + vc * 16));
        }
#pragma unroll
        for (int d = 0; d < DIMS; ++d) s_o[tid][d] = o[d];
        __syncthreads();
        if (wid == 0) {                           // lane = head dim: chunk lane / DIMS of every (wave, row-in-block) partial sum
            const int ch = lane / DIMS, d = lane % DIMS;
            // (the partial sums are READ first, all in flight, and added afterwards in (wave, row) order: csrc/gemv_chain.hip, profiles/r5u_*)
            float part[SELF_WAVES][VROWS];
#pragma unroll
            for (int w = 0; w < SELF_WAVES; ++w)
#pragma unroll
                for (int r = 0; r < VROWS; ++r) part[w][r] = s_o[w * 64 + r * NCH + ch][d];
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < SELF_WAVES; ++w)
#pragma unroll
                for (int r = 0; r < VROWS; ++r) acc += part[w][r];
            for (int j = T; j < nk; ++j) acc = fmaf(s_p[j], (float)s_vnew[j - T][lane], acc);
            p.out[(size_t)(b * p.L + i) * p.ldo + h * 64 + lane] = (h16)f32_as_is(acc);
        }
        __syncthreads();
    }
}

int launch_attn_self(const AttnSelfParams& p, hipStream_t stream) {
    WM_REQUIRE(p.L >= 1 && p.L <= MAX_L, "attn_self: L=%d out of range [1,%d]", p.L, MAX_L);
    WM_REQUIRE(p.T + p.L <= 512, "attn_self: T+L=%d exceeds 512", p.T + p.L);
    WM_REQUIRE(p.T + p.L <= p.present_cap, "attn_self: present capacity %d < T+L=%d", p.present_cap, p.T + p.L);
    WM_REQUIRE(!p.int8_kv || p.kv_scale > 0.f, "attn_self: int8 KV needs a positive scale");
    WM_REQUIRE(p.waves == 0 || p.waves == 1 || p.waves == SELF_WAVES, "attn_self: waves=%d (0, 1 or %d)", p.waves, SELF_WAVES);
    if (p.waves == SELF_WAVES) {
        if (p.int8_kv) hipLaunchKernelGGL(attn_self_wg_kernel<true>, dim3(p.H, p.B), dim3(64 * SELF_WAVES), 0, stream, p);
        else hipLaunchKernelGGL(attn_self_wg_kernel<false>, dim3(p.H, p.B), dim3(64 * SELF_WAVES), 0, stream, p);
    } else if (p.int8_kv)
        hipLaunchKernelGGL(attn_self_kernel<true>, dim3(p.H, p.B), dim3(64), 0, stream, p);
    else
        hipLaunchKernelGGL(attn_self_kernel<false>, dim3(p.H, p.B), dim3(64), 0, stream, p);
    WM_LAUNCH_CHECK(stream, "attn_self");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// cross-attention
// ------------------------------------------------------------------------------------------------
// 256 threads = 4 waves; wave w, pass c covers keys [c*32*4 + w*32 ... ) in groups of 8 rows per
// load instruction (lane -> row lane>>3, 16-byte column lane&7), 4 instructions (4 KiB) in flight per wave.
constexpr int CROSS_MAX_KEYS = 1536;
constexpr int CROSS_MAX_SPLIT = 16;                   // key-range splits the merge kernel combines (the engine uses <= 8)


// I8 (opt-in, beyond the reference: SURVEY 8f-4): K/V are int8 codes [B,2,H,Tk,64] with one scale t per layer; a row is
// 64 B, a lane takes 16 dims (one 16-byte load), a wave-instruction covers 16 rows.  The values are exactly code * t (no
// fp16 rounding of the dequantised tensor), so the scale factors out of both products: score = r16((q16 . code) * t *
// d^-0.25) with v_dot2 on exact fp16 codes, out = r16(t * (p . code)) -- one VALU op per element instead of eight.
// SKIP (round 4; fp16 K/V, single pass): EXACT V-row skipping.  After the two-pass softmax a probability is an fp16 value, and
// with real weights Whisper's cross-attention is sharply peaked: most keys' probabilities round to fp16 ZERO (exp(s - m) / sum <
// 2^-25).  Such a key contributes exactly 0 to P.V (0 x finite = 0, adding 0 changes nothing), so its V row need not leave HBM.
// The load instructions stay where they are -- the software pipeline above depends on an unconditional stream -- but a wave
// instruction whose 8 rows ALL weigh zero (for every token of the call) fetches the item's first 8 V rows instead, which the
// workgroup has just read (an L2 hit): the data is multiplied by 0 either way.  Bit-identical outputs by construction (tests:
// test_cross_attention_v_skip_*); the first block of V rows is requested before the softmax and is never skipped.  With random
// weights (diffuse attention) nothing underflows and nothing is skipped: the bench headline cannot show this, a peaked
// synthetic fixture and FETCH_SIZE do (profiles/r4*_pmc_vskip*).
template <int L, bool I8 = false, int UNR_ = 0, bool SKIP = false>
__global__ __launch_bounds__(256) void attn_cross_kernel(AttnCrossParams p) {
    constexpr int DPL = I8 ? 16 : 8;                     // dims per lane
    constexpr int LPR = 64 / DPL;                        // lanes per row: 8 (fp16) / 4 (int8)
    constexpr int RPI = 64 / LPR;                        // rows per wave-instruction: 8 / 16
    constexpr int ESZ = I8 ? 1 : 2;                      // bytes per stored element
#ifndef CROSS_UNR
#define CROSS_UNR 4
#endif
    // loads per block; two blocks are in flight (see `pipeline`).  2 / 3 / 4 / 6: 6.52 / 6.74 / 6.88 / 6.46 TB/s alone,
    // 24.9 / 25.1 / 25.4 / 26.4 ms per decode step at B = 576 (a gentler stream costs the other groups' chains less)
    constexpr int UNR = UNR_ > 0 ? UNR_ : CROSS_UNR;
    __shared__ float s_sc[L][CROSS_MAX_KEYS];
    __shared__ float s_red[L][4][2];
    __shared__ float s_o[4][L][64];
    __shared__ unsigned char s_live[SKIP ? CROSS_MAX_KEYS / 8 : 1];      // per group of 8 rows (one wave instruction): does any of them weigh anything?

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar loop control
    const int sub = lane % LPR, rowi = lane / LPR;        // 16-byte column, row inside a group of RPI rows
    const int per_split = (((p.Tk + p.nsplit - 1) / p.nsplit) + 7) & ~7;
    constexpr int STRIDE = 4 * RPI * UNR;                 // rows per iteration of the workgroup
    const int first = wid * (RPI * UNR);                  // this wave's first row of an item
    // a workgroup walks over (head, utterance, key split) items: with fewer workgroups than items the
    // launch is persistent and leaves wave slots on every CU to the other streams' short kernels
    // rows still decoding (p.live: count, then indices): finished utterances drop out of the stream, the items are dealt
    // over the live rows only -- the launch's cost follows the live rows (the reference stops at EOT, W/decoding.py:817-819)
    const int n_rows = p.live ? p.live[0] : p.B;
    const int n_items = p.H * n_rows * p.nsplit;

    struct Item { int h, b, sp, k_begin, nkeys; const unsigned char* K; const unsigned char* V; };
    auto geometry = [&](int item) {
        Item it;
        it.h = item % p.H; it.b = (item / p.H) % n_rows; it.sp = item / (p.H * n_rows);
        if (p.live) it.b = p.live[1 + it.b];
        it.k_begin = it.sp * per_split;
        it.nkeys = max(0, min(p.Tk, it.k_begin + per_split) - it.k_begin);
        it.K = (const unsigned char*)p.kv + ((size_t)it.b * p.kv_bstride + ((size_t)(0 * p.H + it.h) * p.Tk) * 64) * ESZ;
        it.V = (const unsigned char*)p.kv + ((size_t)it.b * p.kv_bstride + ((size_t)(1 * p.H + it.h) * p.Tk) * 64) * ESZ;
        return it;
    };
    // UNR 16-byte loads of this wave's rows r0 .. of one K or V matrix (rows past the end re-read the last row)
    auto issue = [&](u32x4 (&dst)[UNR], const unsigned char* base, int k_begin, int nkeys, int r0) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int rr = min(r0 + u * RPI + rowi, nkeys - 1);
            // non-temporal: with plain loads the kernel reads 5.9 instead of 6.6 TB/s alone and the decode step takes 27.3 instead of 25.3 ms
            dst[u] = __builtin_nontemporal_load((const u32x4*)(base + ((size_t)(k_begin + rr) * 64 + sub * DPL) * ESZ));
        }
        __builtin_amdgcn_sched_barrier(0);     // the requests go out BEFORE the work on the rows already here (the scheduler hoists that work otherwise)
    };

    // Software pipeline inside each pass: while a wave works on one block of rows (UNR loads), the next block is in flight
    // in a second register buffer (the buffers swap roles, a copy would have to wait for the prefetch it copies).  The first
    // K block is requested before the q prologue and the first V block before the softmax, so a workgroup meets the HBM
    // latency ~3 times per item instead of 24 times -- a persistent launch has only 8 waves per CU to hide it with:
    // 142 us instead of 174 per launch of 128 utterances (6.9 TB/s).  Nothing is carried across items: that version
    // (scripts/lab/attn_cross_pipelined.patch) needs 175 VGPRs instead of 92 and is no faster.
    auto pipeline = [&](u32x4 (&X)[UNR], u32x4 (&Y)[UNR], int nb, auto&& fetch, auto&& consume) {
        // X holds block 0.  No exit test between the halves of a pair and no branch around a fetch: either leaves a path
        // on which a buffer "may be pending" and the compiler drains the stream at the loop head.
        int k = 0;
        for (; k + 2 < nb; k += 2) {
            fetch(Y, k + 1);
            consume(X, k);
            fetch(X, k + 2);
            consume(Y, k + 1);
        }
        if (nb - k == 2) {
            fetch(Y, k + 1);
            consume(X, k);
            consume(Y, k + 1);
        } else {
            consume(X, k);
        }
    };
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const Item it = geometry(item);
    const int h = it.h, b = it.b, sp = it.sp, k_begin = it.k_begin, nkeys = it.nkeys;
    if (nkeys == 0) {          // empty split (only possible when nsplit > 1): neutral element
        for (int idx = tid; idx < L * 66; idx += 256) {
            float* w = p.ws + ((((size_t)b * p.H + h) * p.nsplit + sp) * L) * 66;
            w[idx] = (idx % 66 == 0) ? -INFINITY : 0.f;
        }
        continue;
    }
    const unsigned char* K = it.K;
    const unsigned char* V = it.V;
    const int nb = (nkeys + STRIDE - 1) / STRIDE;         // blocks of this item; every wave takes all of them (rows past the end are clamped, weigh 0)
    u32x4 bufA[UNR], bufB[UNR];
    issue(bufA, K, k_begin, nkeys, first);

    const size_t sstride = p.part_sstride ? (size_t)p.part_sstride : (size_t)p.B * L * p.ldp;
    // ---- q: this lane's DPL dims (sub * DPL .. ) for each of the L tokens -----------------------------
    // split-K slabs + bias; the slab reads of a round are all in flight together (they are L2 hits, ~1 us each:
    // taken two at a time they were 10 % of a persistent workgroup's time per item)
    float qf[L][DPL];
    {
        constexpr int QV = DPL / 4;
        constexpr int UQ = (8 / (L * QV)) >= 2 ? (8 / (L * QV)) : 2;        // slabs per round, even
        const int col0 = h * 64 + sub * DPL;
        float bs[DPL];
        if (p.bias) {
#pragma unroll
            for (int q8 = 0; q8 < DPL / 8; ++q8) {
                const half8v b8 = *(const half8v*)(p.bias + col0 + q8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) bs[q8 * 8 + e] = (float)b8[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < DPL; ++e) bs[e] = 0.f;
        }
        float4 qa[L][QV];
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int q4 = 0; q4 < QV; ++q4) qa[i][q4] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < p.ksplit; s0 += UQ) {
            float4 a[L][QV][UQ];
#pragma unroll
            for (int i = 0; i < L; ++i)
#pragma unroll
                for (int q4 = 0; q4 < QV; ++q4)
#pragma unroll
                    for (int u = 0; u < UQ; ++u) {
                        const int sl = min(s0 + u, p.ksplit - 1);
                        a[i][q4][u] = *(const float4*)(p.part + (size_t)(b * L + i) * p.ldp + col0 + (size_t)sl * sstride + q4 * 4);
                    }
#pragma unroll
            for (int i = 0; i < L; ++i)
#pragma unroll
                for (int q4 = 0; q4 < QV; ++q4)
#pragma unroll
                    for (int u = 0; u < UQ; u += 2) {      // same pairing as ever: qa += (slab s + slab s+1), absent slabs add 0
                        const bool ok0 = s0 + u < p.ksplit, ok1 = s0 + u + 1 < p.ksplit;
                        const float4 x = a[i][q4][u], y = a[i][q4][u + 1];
                        qa[i][q4].x += (ok0 ? x.x : 0.f) + (ok1 ? y.x : 0.f);
                        qa[i][q4].y += (ok0 ? x.y : 0.f) + (ok1 ? y.y : 0.f);
                        qa[i][q4].z += (ok0 ? x.z : 0.f) + (ok1 ? y.z : 0.f);
                        qa[i][q4].w += (ok0 ? x.w : 0.f) + (ok1 ? y.w : 0.f);
                    }
        }
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int q4 = 0; q4 < QV; ++q4) {
                const float qs[4] = {qa[i][q4].x, qa[i][q4].y, qa[i][q4].z, qa[i][q4].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) qf[i][q4 * 4 + e] = r16(r16(qs[e] + bs[q4 * 4 + e]) * ATTN_SCALE);
            }
    }

    // int8 mode: q as fp16 pairs for v_dot2 (the values are fp16-representable by construction), scale factored out
    half2v qh2[L][DPL / 2];
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int e = 0; e < DPL / 2; ++e) qh2[i][e] = half2v{(h16)qf[i][2 * e], (h16)qf[i][2 * e + 1]};
    const float k_scale = p.kv_q8_scale * ATTN_SCALE;

    // the lane's DPL values of one K / V row as fp32 (fp16 storage, or the int8 codes themselves)
    auto unpack = [&](const u32x4& raw, float (&x)[DPL]) {
        if constexpr (I8) {
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                half2v lo, hi;
                cvt_s8x4_f16x4(w[c], lo, hi);
                x[4 * c + 0] = (float)lo[0]; x[4 * c + 1] = (float)lo[1];
                x[4 * c + 2] = (float)hi[0]; x[4 * c + 3] = (float)hi[1];
            }
        } else {
            const half8v hv = __builtin_bit_cast(half8v, raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (float)hv[e];
        }
    };

    // ---- pass 1: scores ---------------------------------------------------------------------------
    // a wave handles RPI * UNR rows per iteration, the 4 waves stride by 4 * RPI * UNR rows
    float mx[L];
#pragma unroll
    for (int i = 0; i < L; ++i) mx[i] = -INFINITY;
    auto fetch1 = [&](u32x4 (&dst)[UNR], int k) { issue(dst, K, k_begin, nkeys, first + k * STRIDE); };
    auto consume1 = [&](const u32x4 (&cur)[UNR], int k) {
        const int r0 = first + k * STRIDE;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = r0 + u * RPI + rowi;
            float accs[L];
            if constexpr (I8) {
                const uint32_t w4[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
#pragma unroll
                for (int i = 0; i < L; ++i) accs[i] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    half2v lo, hi;
                    cvt_s8x4_f16x4(w4[c], lo, hi);                       // codes as exact fp16 pairs
#pragma unroll
                    for (int i = 0; i < L; ++i) {
                        accs[i] = __builtin_amdgcn_fdot2(qh2[i][2 * c], lo, accs[i], false);
                        accs[i] = __builtin_amdgcn_fdot2(qh2[i][2 * c + 1], hi, accs[i], false);
                    }
                }
            } else {
                float ks[DPL];
                unpack(cur[u], ks);
#pragma unroll
                for (int e = 0; e < DPL; ++e) ks[e] = r16(ks[e] * ATTN_SCALE);
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    float acc = 0.f;
#pragma unroll
                    for (int e = 0; e < DPL; ++e) acc = fmaf(qf[i][e], ks[e], acc);          // (written out: gemv_chain.hip runs a copy of this loop that must round the same way)
                    accs[i] = acc;
                }
            }
#pragma unroll
            for (int i = 0; i < L; ++i) {
                // sum over the LPR lanes that share a row: DPP moves inside the ALU (quad swaps, then the mirrored half-row brings
                // the other quad's sum) instead of __shfl_xor, which compiles to ds_bpermute + a wait on the LDS counter per step --
                // three dependent LDS round trips per row were most of this loop's issue time.  Same pairs, same sums.
                float acc = accs[i];
                acc += wave_dpp<0xB1>(acc);                       // lane ^ 1
                acc += wave_dpp<0x4E>(acc);                       // lane ^ 2
                if constexpr (LPR == 8) acc += wave_dpp<0x141>(acc);      // the other quad of the 8 (row_half_mirror)
                const float sc = I8 ? r16(acc * k_scale) : r16(f32_as_is(acc));
                if (r < nkeys) {
                    if (sub == 0) s_sc[i][r] = sc;
                    mx[i] = fmaxf(mx[i], sc);
                }
            }
        }
    };
    pipeline(bufA, bufB, nb, fetch1, consume1);
    issue(bufA, V, k_begin, nkeys, first);               // first V block: in flight across the softmax
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float m = wave_max_nomfma(mx[i]);
        if (lane == 0) s_red[i][wid][0] = m;
    }
    __syncthreads();
    float gmax[L], gsum[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        gmax[i] = fmaxf(fmaxf(s_red[i][0][0], s_red[i][1][0]), fmaxf(s_red[i][2][0], s_red[i][3][0]));
        float s = 0.f;
        for (int j = tid; j < nkeys; j += 256) {
            const float e = __expf(s_sc[i][j] - gmax[i]);
            s_sc[i][j] = e;
            s += e;
        }
        s = wave_sum_nomfma(s);
        if (lane == 0) s_red[i][wid][1] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < L; ++i) gsum[i] = s_red[i][0][1] + s_red[i][1][1] + s_red[i][2][1] + s_red[i][3][1];
    const bool single = (p.nsplit == 1);
    if (single) {       // exact two-pass softmax: normalise and round the probabilities to fp16
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const float inv = 1.0f / gsum[i];
            for (int j = tid; j < nkeys; j += 256) s_sc[i][j] = r16(s_sc[i][j] * inv);
        }
        __syncthreads();
        if constexpr (SKIP) {
            static_assert(!I8 && RPI == 8, "V-row skipping is built for the fp16 K/V layout (8 rows per wave instruction)");
            for (int g = tid; g * RPI < nkeys; g += 256) {
                bool any = false;
#pragma unroll
                for (int i = 0; i < L; ++i)
#pragma unroll
                    for (int r = 0; r < RPI; ++r) any |= (g * RPI + r < nkeys) && s_sc[i][min(g * RPI + r, nkeys - 1)] != 0.f;
                s_live[g] = any;
            }
            __syncthreads();
        }
    }

    // ---- pass 2: P.V --------------------------------------------------------------------------------
    float o[L][DPL];
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[i][e] = 0.f;
    auto fetch2 = [&](u32x4 (&dst)[UNR], int k) {
        if constexpr (SKIP) {                                     // (the launcher instantiates SKIP for the single-pass form only)
            const int r0 = first + k * STRIDE;
            // the flags of the block's UNR instructions first (wave-uniform LDS reads), then the UNR requests back to back
            bool live[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)     // the instruction's 8 rows r0 + 8 u .. + 7 (r0 is a multiple of 8); groups past the end weigh 0 anyway
                live[u] = (r0 + u * RPI < nkeys) && s_live[min((r0 + u * RPI) / RPI, (nkeys - 1) / RPI)];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int rr = live[u] ? min(r0 + u * RPI + rowi, nkeys - 1) : min(rowi, nkeys - 1);
                dst[u] = __builtin_nontemporal_load((const u32x4*)(V + ((size_t)(k_begin + rr) * 64 + sub * DPL) * ESZ));
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            issue(dst, V, k_begin, nkeys, first + k * STRIDE);
        }
    };
    auto consume2 = [&](const u32x4 (&cur)[UNR], int k) {
        const int r0 = first + k * STRIDE;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = r0 + u * RPI + rowi;
            // no branch around the use of cur[u]: a path that skips it leaves its load pending at the loop head, and the
            // next prefetch into those registers would have to wait for it (rows past the end weigh 0; their data is the last row's)
            float vx[DPL];
            unpack(cur[u], vx);
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const float pr = r < nkeys ? s_sc[i][min(r, nkeys - 1)] : 0.f;
#pragma unroll
                for (int e = 0; e < DPL; ++e) o[i][e] = fmaf(pr, vx[e], o[i][e]);
            }
        }
        // pin the accumulators here: pure arithmetic is free to sink below the NEXT prefetch otherwise, which turns the
        // pipeline into "request two groups, then wait for both"
#pragma unroll
        for (int i = 0; i < L; ++i) {
#pragma unroll
            for (int e = 0; e < DPL; ++e) asm volatile("" : "+v"(o[i][e]) : : "memory");
        }
    };
    pipeline(bufA, bufB, nb, fetch2, consume2);
    // reduce over the row-lanes sharing a column group (lane bits above log2(LPR)), then over the 4 waves
    static_assert(DPL % 8 == 0, "the cross-row steps take eight values at a time");
#pragma unroll
    for (int i = 0; i < L; ++i) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            float v = o[i][e];
            if constexpr (LPR == 4) v += __shfl_xor(v, 4);          // (int8 K/V only: no ALU-side exchange reaches lane ^ 4 exactly)
            v += wave_dpp<0x128>(v);                                // lane ^ 8: row_ror:8 inside a row of 16 lanes
            o[i][e] = v;
        }
#pragma unroll
        for (int e0 = 0; e0 < DPL; e0 += 8) {                       // lane ^ 16, lane ^ 32: permlane swaps, eight values per step (common.h:
            wave_add_xor16_x8_nomfma(&o[i][e0]);                    // no LDS round trips; the same pairs and order as the __shfl_xor loop
            wave_add_xor32_x8_nomfma(&o[i][e0]);                    // this replaces -- and as the single steps did until round 5)
        }
    }
    if (rowi == 0) {
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int e = 0; e < DPL; ++e) s_o[wid][i][sub * DPL + e] = I8 ? o[i][e] * p.kv_q8_scale : o[i][e];
    }
    __syncthreads();
    for (int idx = tid; idx < L * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63;
        const float v = s_o[0][i][d] + s_o[1][i][d] + s_o[2][i][d] + s_o[3][i][d];
        if (single) {
            p.out[(size_t)(b * L + i) * p.ldo + h * 64 + d] = (h16)v;
        } else {
            float* w = p.ws + ((((size_t)b * p.H + h) * p.nsplit + sp) * L + i) * 66;
            w[2 + d] = v;
            if (d == 0) { w[0] = gmax[i]; w[1] = gsum[i]; }
        }
    }
    __syncthreads();          // the next item re-uses the LDS buffers
    }
}

// combine the key-range splits: softmax-weighted merge of (max, sum, unnormalised o).  All of an item's partial results
// are requested at once (lanes < nsplit: max and sum; every lane: its dim of each split's o), the factors travel by
// shuffle -- one memory round trip instead of 3 x nsplit dependent ones (8 -> 2 us at batch 1, where this kernel runs
// 32 times per token).  Sums are taken in split order, as before.
__global__ __launch_bounds__(64) void attn_cross_combine_kernel(AttnCrossParams p) {
    const int h = blockIdx.x, i = blockIdx.z, d = threadIdx.x;
    int b = blockIdx.y;
    if (p.live) {
        if (b >= p.live[0]) return;
        b = p.live[1 + b];
    }
    const float* w = p.ws + ((((size_t)b * p.H + h) * p.nsplit) * p.L + i) * 66;
    const size_t stride = (size_t)p.L * 66;
    float ms = w[min(d, p.nsplit - 1) * stride], ls = w[min(d, p.nsplit - 1) * stride + 1];
    if (d >= p.nsplit) { ms = -INFINITY; ls = 0.f; }
    float ov[CROSS_MAX_SPLIT];                    // (no per-element test around a load; splits past the end re-read the last one, weight 0)
#pragma unroll
    for (int s = 0; s < CROSS_MAX_SPLIT; ++s) ov[s] = w[min(s, p.nsplit - 1) * stride + 2 + d];
    const float m = wave_max_nomfma(ms);
    const float f = d < p.nsplit ? __expf(ms - m) : 0.f;
    const float lf = ls * f;
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int s = 0; s < CROSS_MAX_SPLIT; ++s) {
        if (s < p.nsplit) {                       // wave-uniform
            den += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lf), s));     // (v_readlane: s is a constant after unrolling)
            num += mul_rn(ov[s], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), s)));     // (product and sum rounded separately, as this loop always compiled: common.h)
        }
    }
    p.out[(size_t)(b * p.L + i) * p.ldo + h * 64 + d] = (h16)(num / den);
}

// lab (WM_CROSS_UNR=2): the single-token fp16 kernel with 2 loads per block instead of 4: 78 registers (80 allocated) instead of 92 (96)
static bool cross_unr2() {
    static const bool v = lab_env_int("WM_CROSS_UNR", 4) == 2;
    return v;
}

int launch_attn_cross(const AttnCrossParams& p, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    WM_REQUIRE(p.L >= 1 && p.L <= MAX_L, "attn_cross: L=%d out of range [1,%d]", p.L, MAX_L);
    WM_REQUIRE(p.Tk >= 1 && p.Tk <= CROSS_MAX_KEYS, "attn_cross: Tk=%d out of range", p.Tk);
    WM_REQUIRE(p.nsplit >= 1 && (p.nsplit == 1 || p.ws != nullptr), "attn_cross: split needs a workspace");
    WM_REQUIRE(p.nsplit <= CROSS_MAX_SPLIT, "attn_cross: nsplit=%d exceeds %d (the merge kernel holds one partial result per register)",
               p.nsplit, CROSS_MAX_SPLIT);
    // Persistent launch for big batches: two workgroups per CU walk over the (head, utterance, split) items instead of
    // one workgroup per item.  The kernel alone is as fast either way (6.3-6.7 TB/s), but with 8 of a CU's 32 wave slots
    // it leaves room for the OTHER utterance group's short kernels to be dispatched while it streams: 13.2 instead of
    // 13.9 ms per decode step at B = 256 (WM_CROSS_PERSIST_WGS overrides the workgroup count, 0 = one per item).
    static const int persist_env = lab_env_int("WM_CROSS_PERSIST_WGS", -1);
    // CU count of the device this launch goes to, cached per device (a process may drive several GPUs from several threads)
    static std::atomic<int> n_cu_dev[64];
    int dev = 0;
    WM_CHECK_HIP(hipGetDevice(&dev));
    const int slot = (dev >= 0 && dev < 64) ? dev : 0;
    int n_cu = n_cu_dev[slot].load(std::memory_order_relaxed);
    if (n_cu == 0) {
        int v = 0;
        WM_CHECK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu = v > 0 ? v : 256;
        n_cu_dev[slot].store(n_cu, std::memory_order_relaxed);
    }
    const int n_items = p.H * p.B * p.nsplit;
    // default: at most 2 workgroups per CU, and every workgroup the same number of items (3840 items on 512 workgroups
    // would be 8 for some and 7 for the rest: the launch ends with half the chip idle; 480 x 8 does not).
    // Round 3 re-measured the count with the row-split chain (WM_CROSS_PERSIST_WGS, profiles/r3n_ab_kv_wgs_*.json): 480 / 384 / 320 /
    // 240 workgroups -> token step 23.5 / 23.8 / 23.8 / 23.2 ms at B = 576, chain between two launches 284 / 249 / 208 / 223 us,
    // launches 434 / 479 / 516 / 490 us: with fewer workgroups the chains' round trips queue behind fewer requests and the
    // launches take that much longer -- the sum is what every schedule conserves.  Alone the launch reads 6.6 TB/s with two
    // workgroups per CU and 5.8 with one (252.9 vs 222.7 us, profiles/r3q_bench_kernel_stats.csv), so two it stays.
    int persist_wgs = 0;
    if (persist_env >= 0) persist_wgs = persist_env;
    else if (n_items >= 4 * n_cu) {
        const int per = (n_items + 2 * n_cu - 1) / (2 * n_cu);
        persist_wgs = (n_items + per - 1) / per;
    }
    dim3 grid(persist_wgs > 0 && persist_wgs < n_items ? persist_wgs : n_items);
    if (p.kv_q8_scale > 0.f) {                                   // int8 cross K/V (opt-in)
        switch (p.L) {
            case 1:
                if (ev_start && ev_stop) hipExtLaunchKernelGGL((attn_cross_kernel<1, true>), grid, dim3(256), 0, stream, ev_start, ev_stop, 0, p);
                else hipLaunchKernelGGL((attn_cross_kernel<1, true>), grid, dim3(256), 0, stream, p);
                break;
            case 2: hipLaunchKernelGGL((attn_cross_kernel<2, true>), grid, dim3(256), 0, stream, p); break;
            case 3: hipLaunchKernelGGL((attn_cross_kernel<3, true>), grid, dim3(256), 0, stream, p); break;
            default: hipLaunchKernelGGL((attn_cross_kernel<4, true>), grid, dim3(256), 0, stream, p); break;
        }
    } else if (p.skip_zero_rows && p.nsplit == 1 && !cross_unr2()) {            // exact V-row skipping (fp16 K/V, single pass)
        switch (p.L) {
            case 1:
                if (ev_start && ev_stop) hipExtLaunchKernelGGL((attn_cross_kernel<1, false, 0, true>), grid, dim3(256), 0, stream, ev_start, ev_stop, 0, p);
                else hipLaunchKernelGGL((attn_cross_kernel<1, false, 0, true>), grid, dim3(256), 0, stream, p);
                break;
            case 2: hipLaunchKernelGGL((attn_cross_kernel<2, false, 0, true>), grid, dim3(256), 0, stream, p); break;
            case 3: hipLaunchKernelGGL((attn_cross_kernel<3, false, 0, true>), grid, dim3(256), 0, stream, p); break;
            default: hipLaunchKernelGGL((attn_cross_kernel<4, false, 0, true>), grid, dim3(256), 0, stream, p); break;
        }
    } else if (ev_start && ev_stop && p.L == 1) {
        // in-situ roofline sample (bench.py): the events take the dispatch's own begin / end timestamps, as a
        // profiler would -- events recorded around an ordinary launch add ~45 us of marker latency to a 148 us kernel
        if (cross_unr2()) hipExtLaunchKernelGGL((attn_cross_kernel<1, false, 2>), grid, dim3(256), 0, stream, ev_start, ev_stop, 0, p);
        else hipExtLaunchKernelGGL(attn_cross_kernel<1>, grid, dim3(256), 0, stream, ev_start, ev_stop, 0, p);
    } else switch (p.L) {
        case 1:
            if (cross_unr2()) hipLaunchKernelGGL((attn_cross_kernel<1, false, 2>), grid, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL(attn_cross_kernel<1>, grid, dim3(256), 0, stream, p);
            break;
        case 2: hipLaunchKernelGGL(attn_cross_kernel<2>, grid, dim3(256), 0, stream, p); break;
        case 3: hipLaunchKernelGGL(attn_cross_kernel<3>, grid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL(attn_cross_kernel<4>, grid, dim3(256), 0, stream, p); break;
    }
    WM_LAUNCH_CHECK(stream, "attn_cross");
    if (p.nsplit > 1) {
        hipLaunchKernelGGL(attn_cross_combine_kernel, dim3(p.H, p.B, p.L), dim3(64), 0, stream, p);
        WM_LAUNCH_CHECK(stream, "attn_cross_combine");
    }
    return 0;
}

}  // namespace wm
