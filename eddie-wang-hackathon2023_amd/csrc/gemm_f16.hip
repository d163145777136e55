// fp16 x fp16 -> fp16 GEMM for the compute-bound stages (encoder blocks, convolutions as strided
// views, cross-K/V projection), M = 1500 * batch rows:   C = epilogue(A[M,K] . W[N,K]^T)
//
// Weight-only-int8 engines reach this kernel too: their encoder / cross-K/V weights are expanded
// once, at engine creation, to fp16(fp16(q) * scale) -- exactly the per-element dequantisation the
// reference kernels apply before the multiply (weightOnlyMatrixVectorMultiplication.cu:44-53,
// fpA_intB CUTLASS converter) -- because at M >> 16 the GEMM is MFMA-bound, weight bytes are
// irrelevant, and 288 GB of HBM make the fp16 copy free.
//
// gfx950 structure: 256 x 128 x 64 workgroup tile, 8 waves (4 x 2, each 64 x 64 = 4 x 4 MFMA
// 16x16x32 blocks).  Operand tiles go global -> LDS directly (global_load_lds, 16 B per lane,
// 1 KiB per wave instruction) through a 3-stage ring; a wave waits only for its own loads of the
// tile it is about to read (counted s_waitcnt vmcnt(6): the 6 loads of the next tile stay in
// flight across the barrier) and the workgroup meets at ONE raw s_barrier per K-tile.  LDS rows
// are 128 B; the 16-byte chunk c of row r is stored at position c ^ ((r >> 1) & 7) -- applied on
// the per-lane SOURCE address, since the DMA writes LDS linearly -- which makes every
// ds_read_b128 fragment read (16 rows x one chunk) hit 64 distinct banks.
#include "common.h"
#include "kernels.h"

namespace wm {

namespace f16gemm {
constexpr int BM = 256, BN = 128, BK = 64, STAGES = 3;
constexpr int A_STAGE = BM * BK * 2;            // 32 KiB
constexpr int B_STAGE = BN * BK * 2;            // 16 KiB
constexpr int STAGE = A_STAGE + B_STAGE;        // 48 KiB
constexpr int LOADS = 6;                        // wave-wide 1 KiB loads per wave per K-tile (4 A + 2 B)
}  // namespace f16gemm

__global__ __launch_bounds__(512) void gemm_f16_kernel(GemmBigParams p) {
    using namespace f16gemm;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    const int nwg = gridDim.x, nt_n = p.N / BN;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap: each XCD walks a contiguous band of tiles
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / nt_n, tn = bid % nt_n;
    const int row0 = tm * BM, col0 = tn * BN;

    // ---- loader: wave-instruction i covers tile rows 8i .. 8i+7 (A: i < 32, W: i >= 32) ------------
    const h16* src[LOADS];
    int dst[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int i = wid + 8 * j;
        const bool isA = i < 32;
        const int r = (isA ? i : i - 32) * 8 + (lane >> 3);       // row inside the tile
        const int c = (lane & 7) ^ ((r >> 1) & 7);                 // source chunk for this LDS slot
        if (isA) {
            int gr = row0 + r;
            if (gr > p.M - 1) gr = p.M - 1;
            const size_t off = p.a_rows > 0 ? (size_t)(gr / p.a_rows) * p.a_bstride + (size_t)(gr % p.a_rows) * p.lda
                                            : (size_t)gr * p.lda;
            src[j] = p.A + off + c * 8;
        } else {
            src[j] = (const h16*)p.W + (size_t)(col0 + r) * p.K + c * 8;
        }
        dst[j] = (isA ? 0 : A_STAGE) + (isA ? i : i - 32) * 1024;
    }
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + dst[j]),
                                             16, 0, 0);
    };

    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);

    const int swz = (lane & 15) >> 1, g = lane >> 4;
    const int a_off = (wr * 64 + (lane & 15)) * 128;
    const int b_off = A_STAGE + (wc * 64 + (lane & 15)) * 128;

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < nk) issue(kt + 2, (kt + 2) % STAGES);
        const unsigned char* st = smem + (kt % STAGES) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pos = ((4 * s + g) ^ swz) * 16;
            half8v af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const half8v*)(st + a_off + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *(const half8v*)(st + b_off + j * 16 * 128 + pos);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------
    // The wave's 64 x 64 fp32 tile goes through LDS (the operand ring is free now) so that each lane
    // finishes 8 CONTIGUOUS columns of one row: bias / activation / residual on vectors, one 16-byte
    // store per 8 outputs (8 lanes cover a 128-byte row segment), row addressing computed once per row.
    __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring
    constexpr int EP_LD = 68;                           // floats per staged row (64 + 4 pad)
    float* ep = (float*)smem + wid * (64 * EP_LD);
    {
        const int lc = lane & 15, lr = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(i * 16 + lr + r) * EP_LD + j * 16 + lc] = acc[i][j][r];
    }
    // LDS traffic of one wave only: program order + the compiler's lgkmcnt suffice, no barrier needed
    const int cseg = (lane & 7) * 8;                    // first of this lane's 8 columns inside the wave tile
    const int col = col0 + wc * 64 + cseg;
    float bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bi[e] = 0.f;
    if (p.bias) {
        const half8v b8 = *(const half8v*)(p.bias + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) bi[e] = (float)b8[e];
    }
    const bool scale_cols = col < p.colscale_n;         // colscale_n is a multiple of 64
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rl = it * 8 + (lane >> 3);
        const int row = row0 + wr * 64 + rl;
        if (row >= p.M) continue;
        const float4 v0 = *(const float4*)(ep + rl * EP_LD + cseg);
        const float4 v1 = *(const float4*)(ep + rl * EP_LD + cseg + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = r16(v[e] + bi[e]);                                  // the Linear's fp16 output
            if (p.act == 1) t = r16(gelu_erf(t));
            else if (p.act == 2) t = r16(gelu_tanh(t));
            if (scale_cols) t = r16(t * p.colscale);                      // q, k * d^-0.25 (torch_model.py:93-95)
            v[e] = t;
        }
        if (p.residual) {
            const int rr = p.res_mod > 0 ? row % p.res_mod : row;
            const half8v r8 = *(const half8v*)(p.residual + (size_t)rr * p.ldr + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = r16(v[e] + (float)r8[e]);
        }
        size_t off;
        if (p.out_mode == 0) {
            off = p.c_rows > 0 ? (size_t)(row / p.c_rows) * p.c_bstride + (size_t)(row % p.c_rows) * p.ldc + col
                               : (size_t)row * p.ldc + col;
        } else {   // head-split [B, 2, H, T, 64] (whisper/model.py:519); 8 columns stay inside one head
            const int HC = p.hs_H * 64;
            const int kv = p.hs_kv < 0 ? col / HC : p.hs_kv, cc = p.hs_kv < 0 ? col % HC : col;
            const int b = row / p.hs_T, t = row % p.hs_T, h = cc >> 6, d = cc & 63;
            off = ((((size_t)b * 2 + kv) * p.hs_H + h) * p.hs_T + t) * 64 + d;
        }
        half8v o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)v[e];
        *(half8v*)(p.C + off) = o;
    }
}

int launch_gemm_f16(const GemmBigParams& p, hipStream_t stream) {
    using namespace f16gemm;
    WM_REQUIRE(!p.w8, "gemm_f16: int8 weights must be expanded first");
    WM_REQUIRE(p.N % BN == 0, "gemm_f16: N=%d must be a multiple of %d", p.N, BN);
    WM_REQUIRE(p.K % BK == 0, "gemm_f16: K=%d must be a multiple of %d", p.K, BK);
    WM_REQUIRE(p.lda % 8 == 0, "gemm_f16: lda=%d must be a multiple of 8 (16-byte loads)", p.lda);
    WM_REQUIRE(p.M > 0, "gemm_f16: empty M");
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
    const size_t lds = (size_t)STAGES * STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        WM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_f16_kernel, dim3(grid), dim3(512), lds, stream, p);
    WM_LAUNCH_CHECK(stream, "gemm_f16");
    return 0;
}

// ---- one-off expansion of weight-only int8 matrices to fp16(fp16(q) * scale) ------------------------
__global__ void dequant_w8_kernel(const int8_t* q, const h16* scale, h16* out, int N, int K) {
    const size_t total = (size_t)N * K / 8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 8;
        const h16 s = scale[e / K];
        const uint2 w = *(const uint2*)(q + e);
        half2v h[4];
        cvt_s8x4_f16x4(w.x, h[0], h[1]);
        cvt_s8x4_f16x4(w.y, h[2], h[3]);
        half8v o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] = h[j][0] * s; o[2 * j + 1] = h[j][1] * s; }
        *(half8v*)(out + e) = o;
    }
}

int launch_dequant_w8(const int8_t* q, const h16* scale, h16* out, int N, int K, hipStream_t stream) {
    WM_REQUIRE(K % 8 == 0, "dequant_w8: K=%d must be a multiple of 8", K);
    hipLaunchKernelGGL(dequant_w8_kernel, dim3(2048), dim3(256), 0, stream, q, scale, out, N, K);
    WM_LAUNCH_CHECK(stream, "dequant_w8");
    return 0;
}

}  // namespace wm
