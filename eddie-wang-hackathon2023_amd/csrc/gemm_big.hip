// Dense GEMM for the M >> 16 stages (encoder blocks, both convolutions as strided-view GEMMs,
// the cross-K/V projection):   C[M,N] = epilogue( A[M,K] (fp16) x W[N,K]^T (fp16 | int8+scale) )
//
// Replaces, for this path: the TensorRT MatMul/Convolution tactics of the fp16 engines and
// CutlassFpAIntBGemmRunner<half,uint8_t>::gemm for the weight-only engines
// (R/cpp/tensorrt_llm/kernels/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm_template.h:47-140);
// numerics contract = W8A16, fp32 accumulate (default_fpA_intB_traits.h:30-109).
//
// gfx950 design: 128x128x64 workgroup tile, 4 waves (2x2), each wave 64x64 as 4x4 MFMA
// 16x16x32 f16 blocks (fp32 accumulators in AGPR/VGPR), LDS double buffer with one barrier per
// K-tile, register-staged global loads issued one tile ahead, 144-byte LDS rows so that the
// 16 row-lanes of a ds_read_b128 land on distinct banks.  int8 weights are expanded to fp16
// while they are staged (exact: |q| <= 128), the per-channel scale is applied once to the fp32
// accumulator in the epilogue.  Workgroup ids are remapped so that each XCD owns a contiguous
// band of tiles (its L2 then keeps one A row-panel and the W panels it sweeps).
#include "common.h"
#include "kernels.h"

namespace wm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_ROW = 144;                    // bytes per staged row: 64 halves + 16 B pad
constexpr int TILE_BYTES = BM * LDS_ROW;        // 18432

template <bool W8>
__global__ __launch_bounds__(256) void gemm_big_kernel(GemmBigParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // buffer b: A tile at smem + 2b * TILE_BYTES, W tile right behind it
    auto sA = [&](int b) { return smem + b * 2 * TILE_BYTES; };
    auto sB = [&](int b) { return smem + b * 2 * TILE_BYTES + TILE_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;       // wave position in the 2x2 grid

    // XCD-aware bijective remap (blocks b and b+8 share an XCD)
    const int nwg = gridDim.x, nt_n = p.N / BN;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / nt_n, tn = bid % nt_n;
    const int row0 = tm * BM, col0 = tn * BN;

    // ---- staging assignment -----------------------------------------------------------
    // A (and fp16 W): 1024 16-byte chunks per tile, 4 per thread: chunk c -> row c/8, 16B-col c%8
    // int8 W: 512 chunks of 16 int8, 2 per thread: chunk c -> row c/4, 16-element col c%4
    const h16* Ag[4];
    int a_lds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, r = c >> 3, cc = c & 7;
        int gr = row0 + r;
        if (gr > p.M - 1) gr = p.M - 1;          // clamp: tail rows are computed but never stored
        if (p.a_rows > 0)
            Ag[i] = p.A + (size_t)(gr / p.a_rows) * p.a_bstride + (size_t)(gr % p.a_rows) * p.lda + cc * 8;
        else
            Ag[i] = p.A + (size_t)gr * p.lda + cc * 8;
        a_lds[i] = r * LDS_ROW + cc * 16;
    }
    const unsigned char* Wg[4];
    int w_lds[4];
    if (W8) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, r = c >> 2, cc = c & 3;
            Wg[i] = (const unsigned char*)p.W + (size_t)(col0 + r) * p.K + cc * 16;
            w_lds[i] = r * LDS_ROW + cc * 32;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, r = c >> 3, cc = c & 7;
            Wg[i] = (const unsigned char*)p.W + ((size_t)(col0 + r) * p.K + cc * 8) * 2;
            w_lds[i] = r * LDS_ROW + cc * 16;
        }
    }

    uint4 ra[4], rw[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *(const uint4*)(Ag[i] + kt * BK);
        if (W8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) rw[i] = *(const uint4*)(Wg[i] + kt * BK);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) rw[i] = *(const uint4*)(Wg[i] + kt * BK * 2);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(uint4*)(sA(buf) + a_lds[i]) = ra[i];
        if (W8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                half2v h[8];
                cvt_s8x4_f16x4(rw[i].x, h[0], h[1]);
                cvt_s8x4_f16x4(rw[i].y, h[2], h[3]);
                cvt_s8x4_f16x4(rw[i].z, h[4], h[5]);
                cvt_s8x4_f16x4(rw[i].w, h[6], h[7]);
                uint4 o0, o1;
                o0.x = __builtin_bit_cast(uint32_t, h[0]); o0.y = __builtin_bit_cast(uint32_t, h[1]);
                o0.z = __builtin_bit_cast(uint32_t, h[2]); o0.w = __builtin_bit_cast(uint32_t, h[3]);
                o1.x = __builtin_bit_cast(uint32_t, h[4]); o1.y = __builtin_bit_cast(uint32_t, h[5]);
                o1.z = __builtin_bit_cast(uint32_t, h[6]); o1.w = __builtin_bit_cast(uint32_t, h[7]);
                *(uint4*)(sB(buf) + w_lds[i]) = o0;
                *(uint4*)(sB(buf) + w_lds[i] + 16) = o1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) *(uint4*)(sB(buf) + w_lds[i]) = rw[i];
        }
    };

    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frag_off = (lane & 15) * LDS_ROW + (lane >> 4) * 16;   // row (lane&15), k-group lane>>4
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* a_base = sA(cur) + (wr * 64) * LDS_ROW + frag_off;
        const unsigned char* b_base = sB(cur) + (wc * 64) * LDS_ROW + frag_off;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            half8v af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const half8v*)(a_base + i * 16 * LDS_ROW + s * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *(const half8v*)(b_base + j * 16 * LDS_ROW + s * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------
    // C/D map of mfma 16x16: col = lane & 15, row = 4 * (lane >> 4) + reg
    const int lc = lane & 15, lr = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = col0 + wc * 64 + j * 16 + lc;
        const float sc = (W8 && p.scale) ? (float)p.scale[col] : 1.0f;
        const float bi = p.bias ? (float)p.bias[col] : 0.0f;
        const float cs = (col < p.colscale_n) ? p.colscale : 1.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wr * 64 + i * 16 + lr + r;
                if (row >= p.M) continue;
                float v = r16(acc[i][j][r] * sc + bi);          // the Linear's fp16 output
                if (p.act == 1) v = r16(gelu_erf(v));
                else if (p.act == 2) v = r16(gelu_tanh(v));
                if (col < p.colscale_n) v = r16(v * cs);        // q, k * d^-0.25 (torch_model.py:93-95)
                if (p.residual) {
                    const int rr = p.res_mod > 0 ? row % p.res_mod : row;
                    v = r16(v + (float)p.residual[(size_t)rr * p.ldr + col]);
                }
                size_t off;
                if (p.out_mode == 0) {
                    if (p.c_rows > 0)
                        off = (size_t)(row / p.c_rows) * p.c_bstride + (size_t)(row % p.c_rows) * p.ldc + col;
                    else
                        off = (size_t)row * p.ldc + col;
                } else {   // head-split [B, 2, H, T, 64] (whisper/model.py:519); row = b*T + t, col = h*64 + d
                    const int HC = p.hs_H * 64;
                    const int kv = p.hs_kv < 0 ? col / HC : p.hs_kv, cc = p.hs_kv < 0 ? col % HC : col;
                    const int b = row / p.hs_T, t = row % p.hs_T, h = cc >> 6, d = cc & 63;
                    off = ((((size_t)b * 2 + kv) * p.hs_H + h) * p.hs_T + t) * 64 + d;
                }
                p.C[off] = (h16)v;
            }
        }
    }
}

int launch_gemm_big(const GemmBigParams& p, hipStream_t stream) {
    WM_REQUIRE(p.N % BN == 0, "gemm_big: N=%d must be a multiple of %d", p.N, BN);
    WM_REQUIRE(p.K % BK == 0, "gemm_big: K=%d must be a multiple of %d", p.K, BK);
    WM_REQUIRE(p.lda % 8 == 0, "gemm_big: lda=%d must be a multiple of 8 (16-byte loads)", p.lda);
    WM_REQUIRE(p.M > 0, "gemm_big: empty M");
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
    const size_t lds = 4 * TILE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        WM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        WM_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_big_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    if (p.w8)
        hipLaunchKernelGGL(gemm_big_kernel<true>, dim3(grid), dim3(256), lds, stream, p);
    else
        hipLaunchKernelGGL(gemm_big_kernel<false>, dim3(grid), dim3(256), lds, stream, p);
    WM_LAUNCH_CHECK(stream, "gemm_big");
    return 0;
}

}  // namespace wm
