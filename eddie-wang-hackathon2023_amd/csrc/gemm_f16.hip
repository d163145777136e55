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
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "kernels.h"

namespace wm {

namespace f16gemm {
constexpr int BK = 64;
}  // namespace f16gemm

// BN = 256 (N % 256 == 0: every large-v2 shape) or 128.  NWAVE = 8 waves as 4 (M) x 2 (N), a 256-row tile; a wave owns
// 64 x BN/2 outputs = 4 x TN MFMA blocks.
// NWAVE = 4 (round 4, BN = 128 only): 2 x 2 waves on a 128 x 128 tile, 64 KB of LDS, two workgroups per CU -- the form for
// FEW ROWS (one to a few clips: M = 1500 .. ~12 000).  There the 256 x 256 tiles of the persistent kernel are too few for the
// chip (M = 1500, N = 1280: 30 tiles for 256 CUs, each running the whole K alone; the batch-1 encoder spent 10.5 ms at 216
// TFLOP/s).  An output element is the same chain of MFMA 16x16x32 steps over K in every tile shape, and the epilogue is this
// kernel's own: results are bit-identical to the 256 x 256 forms (tests: test_gemm_small_tiles_bit_identical), so a clip's
// encoder output still does not depend on how many clips share the launch.
// What did NOT help at one tile per CU (M = 1500, N = 1280: 120 tiles; profiles/r4b_*): a 4-deep ring with counted waits, and eight
// waves on the 128 x 128 tile -- out / mlp2 stayed at 19.8 / 58 us to the microsecond in all three forms.  A lone tile is bound
// by what ONE CU can take in from beyond its XCD's L2 (~45 GB/s: 120 workgroups x 2.6 MB of A and W panels at K = 5120 = 58 us),
// not by how the workgroup is organised; smaller tiles use more CUs and read proportionally more, and a K split would change the
// order of the fp32 sums (a clip's output must not depend on its batch).
template <int NWAVE, int BN, int ACT>      // ACT: 0 none, 1 erf-GELU, 2 tanh-GELU (compile-time: keeps the epilogue small)
__global__ __launch_bounds__(NWAVE * 64, 2) void gemm_f16_kernel(GemmBigParams p) {
    using namespace f16gemm;
    constexpr int TM = 4, ROWS_W = 16 * TM;                   // a wave owns 64 rows x BN / 2 channels; two-stage ring
    constexpr int BM = NWAVE / 2 * ROWS_W;
    static_assert((BM + BN) / 8 % NWAVE == 0, "DMA pieces must divide evenly over the waves");
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
    float4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    issue(0, 0);
    // the second-dispatched half of the workgroup loses every issue arbitration against its SIMD partner; a static
    // priority for it evens the two out (+3-4 % on the K loop, scripts/lab/gemm_lab.hip)
    if (NWAVE == 8 && wid >= 4) __builtin_amdgcn_s_setprio(1);

    const int swz = (lane & 15) >> 1, g = lane >> 4;
    const int a_off = (wr * ROWS_W + (lane & 15)) * 128;
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
            half8v af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const half8v*)(st + a_off + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const half8v*)(st + b_off + j * 16 * 128 + pos);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    // Operands were swapped, so a lane holds 4 CONSECUTIVE output channels of one token row per block: bias,
    // activation, q-k scaling and the residual are applied on those in registers and go out as 8-byte
    // accesses straight from the accumulator layout (a wave instruction covers 16 rows x 32 B).  Measured
    // (scripts/lab/gemm_lab.hip, bench_gemm.py): cheaper than transposing the tile through LDS for 16-byte
    // row segments -- 8.7 us per 256 x 256 tile instead of 24.7.
    constexpr int WN_COLS = BN / 2;                                 // columns per wave
    const bool scale_cols = p.colscale_n > 0;
    half4v b4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + wc * WN_COLS + j * 16 + g * 4;
        b4[j] = p.bias ? *(const half4v*)(p.bias + col) : half4v{0, 0, 0, 0};
    }
    // every kernel-argument test sits OUTSIDE the element loops: one unrolled pass per optional step, so the
    // epilogue stays a few hundred instructions (with the tests inside, the unrolled body grew to 18 000
    // instructions and 1 100 branches -- more than the instruction cache -- and cost more than the K loop)
    const int colw = col0 + wc * WN_COLS + g * 4;                   // this lane's first column
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = row0 + wr * ROWS_W + i * 16 + (lane & 15);
        const int rowc = row < p.M ? row : p.M - 1;
        half4v r4[TN];
        if (p.residual) {                                           // all of this row's residual loads first
            const h16* rrow = p.residual + (size_t)(p.res_mod > 0 ? rowc % p.res_mod : rowc) * p.ldr + colw;
#pragma unroll
            for (int j = 0; j < TN; ++j) r4[j] = *(const half4v*)(rrow + j * 16);
        }
        float v[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] = r16(acc[i][j][r] + (float)b4[j][r]);      // the Linear's fp16 output
        if (ACT == 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j][r] = r16(gelu_erf(v[j][r]));
        } else if (ACT == 2) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j][r] = r16(gelu_tanh(v[j][r]));
        }
        if (scale_cols) {                                           // q, k * d^-0.25 (torch_model.py:93-95)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float sc = (colw + j * 16 < p.colscale_n) ? p.colscale : 1.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j][r] = r16(v[j][r] * sc);
            }
        }
        if (p.residual) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j][r] += (float)r4[j][r];
        }
        if (row >= p.M) continue;
        if (p.out_mode == 0) {
            h16* crow = p.C + (p.c_rows > 0 ? (size_t)(row / p.c_rows) * p.c_bstride + (size_t)(row % p.c_rows) * p.ldc
                                            : (size_t)row * p.ldc) + colw;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                *(half4v*)(crow + j * 16) = half4v{(h16)v[j][0], (h16)v[j][1], (h16)v[j][2], (h16)v[j][3]};
        } else {       // head-split [B, 2, H, T, 64] (whisper/model.py:519); a lane's 4 channels stay inside one head
            const int HC = p.hs_H * 64;
            const int bb = row / p.hs_T, t = row % p.hs_T;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = colw + j * 16;
                const int kv = p.hs_kv < 0 ? col / HC : p.hs_kv, cc = p.hs_kv < 0 ? col % HC : col;
                const size_t off = ((((size_t)bb * 2 + kv) * p.hs_H + (cc >> 6)) * p.hs_T + t) * 64 + (cc & 63);
                if (p.q8_inv_scale > 0.f) {      // int8 cross K/V (opt-in): the fp16 result, quantised like the self-attention cache
                    char4 q;
                    q.x = (signed char)fminf(127.f, fmaxf(-128.f, rintf(r16(v[j][0]) * p.q8_inv_scale)));
                    q.y = (signed char)fminf(127.f, fmaxf(-128.f, rintf(r16(v[j][1]) * p.q8_inv_scale)));
                    q.z = (signed char)fminf(127.f, fmaxf(-128.f, rintf(r16(v[j][2]) * p.q8_inv_scale)));
                    q.w = (signed char)fminf(127.f, fmaxf(-128.f, rintf(r16(v[j][3]) * p.q8_inv_scale)));
                    *(char4*)((signed char*)p.C + off) = q;
                } else
                *(half4v*)(p.C + off) = half4v{(h16)v[j][0], (h16)v[j][1], (h16)v[j][2], (h16)v[j][3]};
            }
        }
    }
}

// 256 x 256 tiles a launch must have before the persistent kernel takes it; below, 128 x 128 tiles on two workgroups per CU.
// Measured on MI355X (scripts/bench_gemm_small.py, profiles/r4a_*): see DESIGN.md section 4.  wm_set_gemm_small_tiles() moves it
// (tests pin either side; <= 0 restores the default).
static std::atomic<int> g_small_tiles{GEMM_SMALL_TILES_DEFAULT};
void set_gemm_small_tiles(int tiles) { g_small_tiles.store(tiles > 0 ? tiles : (tiles == 0 ? 0 : GEMM_SMALL_TILES_DEFAULT), std::memory_order_relaxed); }
int get_gemm_small_tiles() { return g_small_tiles.load(std::memory_order_relaxed); }

int launch_gemm_f16(const GemmBigParams& p, hipStream_t stream) {
    using namespace f16gemm;
    // every large-v2 shape (N a multiple of 256) takes the persistent kernel of gemm_f16p.hip: same arithmetic, bit for bit
    // (scripts/lab/gemm_lab3.hip compares the two element by element) -- unless the launch has too few 256 x 256 tiles to
    // fill the chip (a few clips): then this file's 128 x 128 form.  A launch on a budget of CUs (max_wgs) is persistent.
    const bool few = p.max_wgs <= 0 && p.N % 128 == 0 &&
                     (long)((p.M + 255) / 256) * ((p.N + 255) / 256) < (long)g_small_tiles.load(std::memory_order_relaxed);
    if (!few && gemm_f16p_supports(p)) return launch_gemm_f16p(p, stream);
    WM_REQUIRE(p.N % 128 == 0, "gemm_f16: N=%d must be a multiple of 128", p.N);
    WM_REQUIRE(p.K % BK == 0, "gemm_f16: K=%d must be a multiple of %d", p.K, BK);
    WM_REQUIRE(p.lda % 8 == 0, "gemm_f16: lda=%d must be a multiple of 8 (16-byte loads)", p.lda);
    WM_REQUIRE(p.ldc % 8 == 0 && p.ldr % 8 == 0, "gemm_f16: ldc/ldr must be multiples of 8 (16-byte epilogue accesses)");
    WM_REQUIRE(p.M > 0, "gemm_f16: empty M");
    WM_REQUIRE(p.act >= 0 && p.act <= 2, "gemm_f16: act=%d", p.act);
    // 0: 256 x 256, 1: 256 x 128, 2: 128 x 128 (4 waves, two workgroups per CU), 3: 64 x 128 (2 waves) when even the 128 x 128 tiles
    // would leave more than a third of the CUs idle (ONE clip's n_state-wide projections: 120 tiles): a lone tile is bound by what
    // its CU ingests, so twice the workgroups with three quarters of the bytes each are faster although they read 1.5 x as much in all
    static const int tiny_max = lab_env_int("WM_GEMM_TINY_TILES", GEMM_TINY_TILES_DEFAULT);
    const long tiles128 = (long)((p.M + 127) / 128) * (p.N / 128);
    const int form = few ? (tiles128 <= tiny_max ? 3 : 2) : (p.N % 256 == 0 ? 0 : 1);
    const int bm = form == 3 ? 64 : (form == 2 ? 128 : 256), bn = form == 0 ? 256 : 128;
    const int grid = ((p.M + bm - 1) / bm) * (p.N / bn);
    static std::atomic<unsigned long long> attr_set{0};             // per device: the dynamic-LDS attribute is a per-device property
    int dev = 0;
    WM_CHECK_HIP(hipGetDevice(&dev));
    const int slot = dev & 63;
    using Kern = void (*)(GemmBigParams);
    static const Kern kerns[4][3] = {{gemm_f16_kernel<8, 256, 0>, gemm_f16_kernel<8, 256, 1>, gemm_f16_kernel<8, 256, 2>},
                                     {gemm_f16_kernel<8, 128, 0>, gemm_f16_kernel<8, 128, 1>, gemm_f16_kernel<8, 128, 2>},
                                     {gemm_f16_kernel<4, 128, 0>, gemm_f16_kernel<4, 128, 1>, gemm_f16_kernel<4, 128, 2>},
                                     {gemm_f16_kernel<2, 128, 0>, gemm_f16_kernel<2, 128, 1>, gemm_f16_kernel<2, 128, 2>}};
    static const int rows_of[4] = {256, 256, 128, 64}, cols_of[4] = {256, 128, 128, 128};
    const unsigned long long bit = 1ull << slot;
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        for (int f = 0; f < 4; ++f)
            for (int a = 0; a < 3; ++a)
                WM_CHECK_HIP(hipFuncSetAttribute((const void*)kerns[f][a], hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 2 * (rows_of[f] + cols_of[f]) * BK * 2));
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    const size_t lds = (size_t)2 * (bm + bn) * BK * 2;              // the two-stage K-tile ring
    hipLaunchKernelGGL(kerns[form][p.act], dim3(grid), dim3(form == 3 ? 128 : (form == 2 ? 256 : 512)), lds, stream, p);
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
