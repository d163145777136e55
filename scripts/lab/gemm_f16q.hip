// Lab kernel (not product code): the encoder GEMM as TWO independent 4-wave workgroups per CU, each walking over 128 x 256
// tiles with its own 3-slot LDS-DMA ring.  The question it answers: does the epilogue of one workgroup hide under the K loop
// of the other (the two waves of a SIMD then belong to different workgroups and are never in the same phase for long), and
// does that pay for 1.5x the L2 -> LDS traffic of the 256 x 256 tile?  Same arithmetic and epilogues as gemm_f16p.hip.
#include <type_traits>

namespace wm {
namespace f16q {
constexpr int BM = 128, BN = 256, BK = 32, NWAVE = 4, STAGES = 3;
constexpr int A_PART = BM * BK * 2, STAGE = (BM + BN) * BK * 2;      // 8 KB + 16 KB
constexpr int N_STORES = 32, PIECES = 6;
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void gemm_f16q_kernel(GemmBigParams p) {
    using namespace f16q;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;           // 2 (M) x 2 (N) waves, each 64 rows x 128 channels
    const int g = lane >> 4;

    const int nt_n = p.N / BN, nt_m = (p.M + BM - 1) / BM, n_tiles = nt_n * nt_m;
    const int per_xcd = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3;
    const int band = (n_tiles + 7) >> 3;
    const int lo = min(n_tiles, xcd * band), hi = min(n_tiles, lo + band);
    const int my_tiles = (hi - lo - j0 + per_xcd - 1) > 0 ? (hi - lo - j0 + per_xcd - 1) / per_xcd : 0;
    if (my_tiles == 0) return;
    const int nk = p.K / BK;
    const int total_stages = my_tiles * nk;

    const unsigned char* a_base = nullptr;
    const unsigned char* w_base = nullptr;
    uint32_t a_lane[2], w_lane[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wid + NWAVE * q) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((4 - ((r >> 2) & 3)) & 3);
        w_lane[q] = (uint32_t)r * (uint32_t)p.K * 2u + c * 16;
    }
    auto set_tile = [&](int t) {
        const int tile = lo + j0 + t * per_xcd;
        const int tm = tile / nt_n, tn = tile - tm * nt_n;
        a_base = (const unsigned char*)(p.A + (size_t)tm * BM * p.lda);
        w_base = (const unsigned char*)p.W + (size_t)tn * BN * p.K * 2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wid + NWAVE * q) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((4 - ((r >> 2) & 3)) & 3);
            int gr = tm * BM + r;
            if (gr > p.M - 1) gr = p.M - 1;
            a_lane[q] = (uint32_t)((size_t)(gr - tm * BM) * p.lda * 2) + c * 16;
        }
    };
    int load_ks = 0, load_tile = 0, issued = 0;
    auto issue_stage = [&]() {
        if (issued >= total_stages) return;
        unsigned char* slot = smem + (issued % STAGES) * STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_lane[q]),
                                             (__attribute__((address_space(3))) void*)(slot + (wid + NWAVE * q) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_base + w_lane[q]),
                                             (__attribute__((address_space(3))) void*)(slot + A_PART + (wid + NWAVE * q) * 1024), 16, 0, 0);
        a_base += BK * 2; w_base += BK * 2;
        ++issued;
        if (++load_ks == nk) { load_ks = 0; ++load_tile; if (load_tile < my_tiles) set_tile(load_tile); }
    };
    set_tile(0);
    issue_stage();
    issue_stage();

    const int sw = (4 - ((lane >> 2) & 3)) & 3;
    const int a_off = (wr * 64 + (lane & 15)) * 64 + ((g ^ sw) << 4);
    const int b_off = A_PART + (wc * 128 + (lane & 15)) * 64 + ((g ^ sw) << 4);

    float4v acc[4][8];
    half8v af[4], bx[8];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    int cons = 0, ks = 0, t = 0;
    int relaxed = 0;                                 // stage waits that must leave the epilogue's stores in flight
    for (;;) {
        const unsigned char* st = smem + (cons % STAGES) * STAGE;
        // stage `cons` has landed: only the next stage's pieces (and, after an epilogue, its stores) may still be in flight
        if (issued >= total_stages) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (relaxed > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES + N_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
        if (relaxed > 0) --relaxed;
        __builtin_amdgcn_s_barrier();                // everybody's pieces of `cons` are in LDS, nobody still reads slot cons - 1
        issue_stage();                               // stage cons + 2 -> the slot stage cons - 1 occupied
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const half8v*)(st + a_off + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) bx[j] = *(const half8v*)(st + b_off + j * 1024);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bx[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        ++cons;
        if (++ks < nk) continue;

        // ================================ epilogue ===============================================================
        const int tile = lo + j0 + t * per_xcd;
        const int tm = tile / nt_n, tn = tile - tm * nt_n;
        const int row0 = tm * BM, col0 = tn * BN;
        int le = lane;
        asm volatile("" : "+v"(le));
        const int ge = le >> 4, rl = le & 15;
        const bool scale_cols = p.colscale_n > 0;
        const int colw = col0 + wc * 128 + ge * 4;
        auto finish = [&](auto res_tag) {
            constexpr bool RES = decltype(res_tag)::value;
            half4v b4[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b4[j] = p.bias ? *(const half4v*)(p.bias + colw + j * 16) : half4v{(h16)0.f, (h16)0.f, (h16)0.f, (h16)0.f};
            half4v r4[RES ? 4 : 1][RES ? 8 : 1];
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + wr * 64 + i * 16 + rl;
                    const int rowc = row < p.M ? row : p.M - 1;
                    const h16* rrow = p.residual + (size_t)rowc * p.ldr + colw;
#pragma unroll
                    for (int j = 0; j < 8; ++j) r4[i][j] = *(const half4v*)(rrow + j * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + wr * 64 + i * 16 + rl;
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const int colp = colw + jh * 64;
                    float v[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[j][r] = r16(acc[i][jh * 4 + j][r] + (float)b4[jh * 4 + j][r]);
                    if (ACT == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[j][r] = r16(gelu_erf(v[j][r]));
                    }
                    if (scale_cols) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float sc = (colp + j * 16 < p.colscale_n) ? p.colscale : 1.0f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[j][r] = r16(v[j][r] * sc);
                        }
                    }
                    if constexpr (RES) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[j][r] += (float)r4[i][jh * 4 + j][r];
                    }
                    if (row < p.M) {
                        h16* crow = p.C + (size_t)row * p.ldc + colp;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *(half4v*)(crow + j * 16) = half4v{(h16)v[j][0], (h16)v[j][1], (h16)v[j][2], (h16)v[j][3]};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (p.residual) finish(std::true_type{}); else finish(std::false_type{});
        if (row0 + BM <= p.M) relaxed = 2; else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ks = 0;
        if (++t == my_tiles) break;
        zero_acc();
    }
}

int launch_gemm_f16q(const GemmBigParams& p, hipStream_t stream, int wgs_per_cu = 2) {
    using namespace f16q;
    if (p.N % BN || p.K % 64 || p.K < 128 || p.a_rows || p.c_rows || p.out_mode || p.res_mod) return 1;
    using Kern = void (*)(GemmBigParams);
    static const Kern kerns[2] = {gemm_f16q_kernel<0>, gemm_f16q_kernel<1>};
    constexpr size_t LDS_BYTES = (size_t)STAGES * STAGE;
    static bool attr = false;
    if (!attr) {
        for (int a = 0; a < 2; ++a) (void)hipFuncSetAttribute((const void*)kerns[a], hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        attr = true;
    }
    const int n_tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
    int grid = 256 * wgs_per_cu;
    const int need = ((n_tiles + 7) / 8) * 8;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(kerns[p.act], dim3(grid), dim3(256), LDS_BYTES, stream, p);
    return 0;
}
}  // namespace wm
