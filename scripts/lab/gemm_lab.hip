// Lab harness (not product code): ablations of the fp16 GEMM main loop on the GPU box.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h16;
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ float fast_erf(float x) {   // Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float y = 1.0f - poly * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float r16(float x) { return (float)(h16)x; }
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WM, int WN, int STAGES, int FLAGS, int EPI = 0>   // FLAGS: 1 no glds in loop, 2 no mfma, 4 no xcd remap; EPI 1: swapped operands + real epilogue (bias, residual, gelu if EPI==2)
__global__ __launch_bounds__(WM * WN * 64) void k(const h16* A, const h16* W, h16* C, int M, int N, int K, const h16* bias = nullptr, const h16* resid = nullptr) {
    constexpr int BK = 64, NW = WM * WN;
    constexpr int A_STAGE = BM * BK * 2, B_STAGE = BN * BK * 2, STAGE = A_STAGE + B_STAGE;
    constexpr int NI = (BM + BN) / 8;                 // wave-instrs per k-tile
    constexpr int LOADS = NI / NW;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;   // MFMA blocks per wave
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    const int nwg = gridDim.x, nt_n = N / BN;
    int bid = blockIdx.x;
    if (!(FLAGS & 4)) { const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int tm = bid / nt_n, tn = bid % nt_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const h16* src[LOADS]; int dst[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int i = wid + NW * j;
        const bool isA = i < BM / 8;
        const int r = (isA ? i : i - BM / 8) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        if (isA) { int gr = row0 + r; if (gr > M - 1) gr = M - 1; src[j] = A + (size_t)gr * K + c * 8; }
        else src[j] = W + (size_t)(col0 + r) * K + c * 8;
        dst[j] = (isA ? 0 : A_STAGE) + (isA ? i : i - BM / 8) * 1024;
    }
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + dst[j]), 16, 0, 0);
    };
    float4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) if (s < nk) issue(s, s);
    const int swz = (lane & 15) >> 1, g = lane >> 4;
    const int a_off = (wr * (BM / WM) + (lane & 15)) * 128;
    const int b_off = A_STAGE + (wc * (BN / WN) + (lane & 15)) * 128;
    if ((FLAGS & 16) && wid >= 4) __builtin_amdgcn_s_setprio(1);
    if ((FLAGS & 32) && wid >= 4) {
        // staggered half of the workgroup: runs half a K-tile behind its SIMD partners, so that one wave's LDS
        // reads fall under the other's MFMAs.  The second half's fragments of K-tile kt are read before the
        // next barrier and multiplied after it.
        half8v af1[TM], bf1[TN];
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < nk) issue(kt + 1, (kt + 1) % STAGES);
            if (kt > 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[i], bf1[j], acc[i][j], 0, 0, 0);
            }
            const unsigned char* st = smem + (kt % STAGES) * STAGE;
            {
                const int pos = ((4 * 0 + g) ^ swz) * 16;
                half8v af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const half8v*)(st + a_off + i * 16 * 128 + pos);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *(const half8v*)(st + b_off + j * 16 * 128 + pos);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            {
                const int pos = ((4 * 1 + g) ^ swz) * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) af1[i] = *(const half8v*)(st + a_off + i * 16 * 128 + pos);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf1[j] = *(const half8v*)(st + b_off + j * 16 * 128 + pos);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[i], bf1[j], acc[i][j], 0, 0, 0);
    } else
    for (int kt = 0; kt < nk; ++kt) {
        if (!(FLAGS & 1)) {
            // tiles kt+1 .. kt+STAGES-2 may stay in flight
            const int inflight = min(STAGES - 2, nk - 1 - kt);
            if (inflight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
            else if (inflight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (kt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!(FLAGS & 1) && kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
        const unsigned char* st = smem + ((FLAGS & 1) ? 0 : (kt % STAGES)) * STAGE;
        if (FLAGS & 8) {
            // fragments of both K-halves up front: the second half's LDS reads fly under the first half's MFMAs
            half8v af[2][TM], bf[2][TN];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pos = ((4 * s + g) ^ swz) * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[s][i] = *(const half8v*)(st + a_off + i * 16 * 128 + pos);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[s][j] = *(const half8v*)(st + b_off + j * 16 * 128 + pos);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int pos = ((4 * s + g) ^ swz) * 16;
            half8v af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const half8v*)(st + a_off + i * 16 * 128 + pos);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const half8v*)(st + b_off + j * 16 * 128 + pos);
            if (FLAGS & 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bf[j]));
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (EPI) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        }
    }
    if (EPI) {   // D = W.A^T blocks: lane holds row m = lane&15 (per i), 4 consecutive n = 4g + r (per j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = row0 + wr * (BM / WM) + i * 16 + (lane & 15);
            if (row >= M) continue;
            h16* crow = C + (size_t)row * N;
            const h16* rrow = resid ? resid + (size_t)row * N : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + wc * (BN / WN) + j * 16 + g * 4;
                const half4v b4 = *(const half4v*)(bias + col);
                half4v o;
                half4v r4 = {0, 0, 0, 0};
                if (rrow) r4 = *(const half4v*)(rrow + col);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = r16(acc[i][j][r] + (float)b4[r]);
                    if (EPI == 2) v = r16(0.5f * v * (1.0f + fast_erf(v * 0.70710678f)));
                    if (rrow) v = r16(v + (float)r4[r]);
                    o[r] = (h16)v;
                }
                *(half4v*)(crow + col) = o;
            }
        }
        return;
    }
    // minimal epilogue: keep results live, write one value per lane per block
    const int lc = lane & 15, lr = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = row0 + wr * (BM / WM) + i * 16 + lr, col = col0 + wc * (BN / WN) + j * 16 + lc;
            if (row < M) C[(size_t)row * N + col] = (h16)(acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]);
        }
}

template <int BM, int BN, int WM, int WN, int STAGES, int FLAGS, int EPI = 0>
void run(const char* name, const h16* A, const h16* W, h16* C, int M, int N, int K, const h16* bias = nullptr, const h16* resid = nullptr) {
    constexpr int lds = STAGES * (BM + BN) * 64 * 2;
    auto kern = k<BM, BN, WM, WN, STAGES, FLAGS, EPI>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = ((M + BM - 1) / BM) * (N / BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, 0, A, W, C, M, N, K, bias, resid);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, 0, A, W, C, M, N, K, bias, resid);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-38s M=%d N=%d K=%d lds=%dK grid=%d: %.3f ms %.0f TF/s\n", name, M, N, K, lds / 1024, grid, ms, 2.0 * M * N * K / ms / 1e9);
}

int main() {
    const int M = 48000;
    const size_t maxA = (size_t)M * 5120, maxW = (size_t)5120 * 5120;
    std::vector<h16> hA(maxA), hW(maxW);
    srand(1);
    for (auto& x : hA) x = (h16)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& x : hW) x = (h16)((rand() % 2001 - 1000) / 30000.0f);
    h16 *A, *W, *C;
    CK(hipMalloc(&A, maxA * 2)); CK(hipMalloc(&W, maxW * 2)); CK(hipMalloc(&C, (size_t)M * 5120 * 2));
    CK(hipMemcpy(A, hA.data(), maxA * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), maxW * 2, hipMemcpyHostToDevice));
    h16 *bias, *R;
    CK(hipMalloc(&bias, 5120 * 2)); CK(hipMemset(bias, 0, 5120 * 2));
    CK(hipMalloc(&R, (size_t)M * 5120 * 2)); CK(hipMemset(R, 0, (size_t)M * 5120 * 2));
    struct S { int N, K; } shapes[] = {{1280, 5120}, {3840, 1280}, {1280, 5120}};
    for (auto sh : shapes) {
        const int N = sh.N, K = sh.K;
        run<256, 256, 4, 2, 2, 16, 0>("256x256 2st setprio waves4-7", A, W, C, M, N, K);
        run<256, 256, 4, 2, 2, 32, 0>("256x256 2st stagger waves4-7", A, W, C, M, N, K);
        run<256, 256, 4, 2, 2, 48, 0>("256x256 2st stagger+setprio", A, W, C, M, N, K);
    }
    return 0;
}
