// fp16 x fp16 -> fp16 GEMM for the compute-bound stages (encoder blocks, convolutions as strided
// views, cross-K/V projection), M = 1500 * batch rows:   C = epilogue(A[M,K] . W[N,K]^T)
//
// Weight-only-int8 engines reach this kernel too: their encoder / cross-K/V weights are expanded
// once, at engine creation, to fp16(fp16(q) * scale) -- exactly the per-element dequantisation the
// reference kernels apply before the multiply (weightOnlyMatrixVectorMultiplication.cu:44-53,
// fpA_intB CUTLASS converter) -- because at M >> 16 the GEMM is MFMA-bound, weight bytes are
// irrelevant, and 288 GB of HBM make the fp16 copy free.
//
// gfx950 structure: 256 x 256 x 64 workgroup tile (256 x 128 when N is not a multiple of 256),
// 8 waves (4 x 2, each 64 x 128 = 4 x 8 MFMA 16x16x32 blocks, 128 accumulator VGPRs).  Operand
// tiles go global -> LDS directly (global_load_lds, 16 B per lane, 1 KiB per wave instruction)
// into a 2-stage ring: tile k+1 is in flight while tile k is multiplied, the workgroup meets at
// ONE raw s_barrier per K-tile.  Measured on MI355X (scripts/lab/gemm_lab.hip, random data, M = 48000):
// 2 stages beat 3 (873 vs 799 TF/s at 256x128) and 256x256 beats 256x128 (1066-1137 vs 863-959
// TF/s before the epilogue): half the operand bytes per flop through L2 and LDS.  The MFMA operands
// are SWAPPED (D = W.A^T), so each lane ends up with 4 consecutive output channels of one token
// row and the epilogue needs no LDS transpose (8-byte vector accesses for bias/residual/store).
// LDS rows
// are 128 B; the 16-byte chunk c of row r is stored at position c ^ ((r >> 1) & 7) -- applied on
// the per-lane SOURCE address, since the DMA writes LDS linearly -- which makes every
// ds_read_b128 fragment read (16 rows x one chunk) hit 64 distinct banks.
#include "common.h"
#include "kernels.h"

namespace wm {

namespace f16gemm {
constexpr int BM = 256, BK = 64, STAGES = 2, NWAVE = 8;
}  // namespace f16gemm

// BN = 256 (N % 256 == 0: every large-v2 shape) or 128.  8 waves as 4 (M) x 2 (N); a wave owns
// 64 x BN/2 outputs = 4 x TN MFMA blocks.
template <int BN>
__global__ __launch_bounds__(512) void gemm_f16_kernel(GemmBigParams p) {
    using namespace f16gemm;
    constexpr int A_STAGE = BM * BK * 2, B_STAGE = BN * BK * 2, STAGE = A_STAGE + B_STAGE;
    constexpr int LOADS = (BM + BN) / 8 / NWAVE;          // wave-wide 1 KiB DMA loads per wave per K-tile
    constexpr int TN = BN / 2 / 16;
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

    // ---- loader: wave-instruction i covers tile rows 8i .. 8i+7 (A: i < BM/8, W: the rest) ----------
    const h16* src[LOADS];
    int dst[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int i = wid + NWAVE * j;
        const bool isA = i < BM / 8;
        const int r = (isA ? i : i - BM / 8) * 8 + (lane >> 3);   // row inside the tile
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
        dst[j] = (isA ? 0 : A_STAGE) + (isA ? i : i - BM / 8) * 1024;
    }
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + dst[j]),
                                             16, 0, 0);
    };

    // acc[i][j] = D block of W_j . A_i^T: rows = 4 output channels (4g + r), col = token row (lane & 15)
    float4v acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    issue(0, 0);

    const int swz = (lane & 15) >> 1, g = lane >> 4;
    const int a_off = (wr * 64 + (lane & 15)) * 128;
    const int b_off = A_STAGE + (wc * (BN / 2) + (lane & 15)) * 128;

    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's part of tile kt has landed
        __builtin_amdgcn_s_barrier();                              // ... everybody's; stage (kt+1)&1 is free
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);              // flies during the MFMAs below
        const unsigned char* st = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pos = ((4 * s + g) ^ swz) * 16;
            half8v af[4], bf[TN];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const half8v*)(st + a_off + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const half8v*)(st + b_off + j * 16 * 128 + pos);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    // Operands were swapped, so a lane holds 4 CONSECUTIVE output channels of one token row per block:
    // bias / activation / q-k scaling are applied on those in registers (fp32 -> the Linear's fp16
    // output), the fp16 tile of the wave (64 x BN/2) is transposed through LDS (the ring is free), and
    // the residual add + store then run on whole 16-byte row segments: all residual loads of a lane
    // are issued before the first store (a store in between would make the counted vmcnt wait for it).
    __builtin_amdgcn_s_barrier();                                   // every wave is done reading the ring
    constexpr int WN_COLS = BN / 2;                                 // columns per wave
    constexpr int EPLD = WN_COLS + 8;                               // halves per staged row (16-byte pad)
    h16* ep = (h16*)smem + (size_t)wid * 64 * EPLD;
    {
        half4v b4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wc * WN_COLS + j * 16 + g * 4;
            b4[j] = p.bias ? *(const half4v*)(p.bias + col) : half4v{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + wc * WN_COLS + j * 16 + g * 4;
                half4v o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = r16(acc[i][j][r] + (float)b4[j][r]);            // the Linear's fp16 output
                    if (p.act == 1) v = r16(gelu_erf(v));
                    else if (p.act == 2) v = r16(gelu_tanh(v));
                    if (col < p.colscale_n) v = r16(v * p.colscale);          // q, k * d^-0.25 (torch_model.py:93-95)
                    o[r] = (h16)v;
                }
                *(half4v*)(ep + (i * 16 + (lane & 15)) * EPLD + j * 16 + g * 4) = o;
            }
        }
    }
    // one wave's private LDS region: program order + lgkmcnt are enough, no barrier
    constexpr int SEG = WN_COLS / 8;                                // 16-byte segments per row (8 or 16)
    constexpr int RPI = 64 / SEG;                                   // rows covered per wave-instruction (8 or 4)
    constexpr int NIT = 64 / RPI;                                   // iterations (8 or 16)
    const int cseg = (lane % SEG) * 8, rsub = lane / SEG;
    const int colw = col0 + wc * WN_COLS + cseg;
    half8v res[NIT];
    if (p.residual) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int row = row0 + wr * 64 + it * RPI + rsub;
            if (row > p.M - 1) row = p.M - 1;
            res[it] = *(const half8v*)(p.residual + (size_t)(p.res_mod > 0 ? row % p.res_mod : row) * p.ldr + colw);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int rl = it * RPI + rsub;
        const int row = row0 + wr * 64 + rl;
        half8v o = *(const half8v*)(ep + rl * EPLD + cseg);
        if (p.residual) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)((float)o[e] + (float)res[it][e]);
        }
        if (row >= p.M) continue;
        size_t off;
        if (p.out_mode == 0) {
            off = p.c_rows > 0 ? (size_t)(row / p.c_rows) * p.c_bstride + (size_t)(row % p.c_rows) * p.ldc + colw
                               : (size_t)row * p.ldc + colw;
        } else {   // head-split [B, 2, H, T, 64] (whisper/model.py:519); 8 channels stay inside one head
            const int HC = p.hs_H * 64;
            const int kv = p.hs_kv < 0 ? colw / HC : p.hs_kv, cc = p.hs_kv < 0 ? colw % HC : colw;
            const int bb = row / p.hs_T, t = row % p.hs_T, h = cc >> 6, d = cc & 63;
            off = ((((size_t)bb * 2 + kv) * p.hs_H + h) * p.hs_T + t) * 64 + d;
        }
        *(half8v*)(p.C + off) = o;
    }
}

int launch_gemm_f16(const GemmBigParams& p, hipStream_t stream) {
    using namespace f16gemm;
    WM_REQUIRE(!p.w8, "gemm_f16: int8 weights must be expanded first");
    WM_REQUIRE(p.N % 128 == 0, "gemm_f16: N=%d must be a multiple of 128", p.N);
    WM_REQUIRE(p.K % BK == 0, "gemm_f16: K=%d must be a multiple of %d", p.K, BK);
    WM_REQUIRE(p.lda % 8 == 0, "gemm_f16: lda=%d must be a multiple of 8 (16-byte loads)", p.lda);
    WM_REQUIRE(p.ldc % 8 == 0 && p.ldr % 8 == 0, "gemm_f16: ldc/ldr must be multiples of 8 (16-byte epilogue accesses)");
    WM_REQUIRE(p.M > 0, "gemm_f16: empty M");
    const bool wide = (p.N % 256 == 0);
    const int bn = wide ? 256 : 128;
    const int grid = ((p.M + BM - 1) / BM) * (p.N / bn);
    // ring, or the epilogue's per-wave fp16 staging tile (8 waves x 64 rows x (bn/2 + 8) halves), whichever is larger
    const size_t ring = (size_t)STAGES * (BM + bn) * BK * 2, stage = (size_t)8 * 64 * (bn / 2 + 8) * 2;
    const size_t lds = ring > stage ? ring : stage;
    static bool attr_set = false;
    if (!attr_set) {
        WM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f16_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         8 * 64 * (128 + 8) * 2 > STAGES * (BM + 256) * BK * 2 ? 8 * 64 * (128 + 8) * 2 : STAGES * (BM + 256) * BK * 2));
        WM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_f16_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         STAGES * (BM + 128) * BK * 2));
        attr_set = true;
    }
    if (wide) hipLaunchKernelGGL(gemm_f16_kernel<256>, dim3(grid), dim3(512), lds, stream, p);
    else hipLaunchKernelGGL(gemm_f16_kernel<128>, dim3(grid), dim3(512), lds, stream, p);
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
