// Lab: issue cadence of v_mfma_f32_16x16x32_f16 with 16 independent accumulators (the persistent GEMM's multiply interval:
// 4 A fragments x 4 B fragments), accumulators in architectural VGPRs (what hipcc emits for gemm_f16p.hip) against accumulators
// in AGPRs (inline assembly), one or two waves per SIMD, on random operands.  Core-clock cycles per MFMA (s_memtime).
//   hipcc -O3 --offload-arch=gfx950 scripts/lab/mfma_issue_lab.hip -o /tmp/mfma_issue_lab && /tmp/mfma_issue_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>      // 0: builtin (VGPR accumulators), 1: inline asm with AGPR accumulators, 2: inline asm with VGPR accumulators
__global__ __launch_bounds__(512) void k(const half8v* in, float* out, unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x & 63;
    half8v a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(blockIdx.x * 8 + i) * 64 + lane]; b[i] = in[(blockIdx.x * 8 + 4 + i) * 64 + lane]; }
    float4v acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 0) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i * 4 + j], 0, 0, 0);
                else if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i * 4 + j]) : "v"(b[j]), "v"(a[i]));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i * 4 + j]) : "v"(b[j]), "v"(a[i]));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    const int n_wg = 256, reps = 2000;
    half8v* in; float* out; unsigned long long* cyc;
    std::vector<_Float16> h(n_wg * 8 * 64 * 8);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    CK(hipMalloc(&in, h.size() * 2)); CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, n_wg * 512 * 4)); CK(hipMalloc(&cyc, n_wg * 8 * 8));
    const char* names[3] = {"builtin (VGPR accumulators)", "asm, AGPR accumulators", "asm, VGPR accumulators"};
    for (int threads : {256, 512})
        for (int mode = 0; mode < 3; ++mode) {
            for (int it = 0; it < 3; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(n_wg), dim3(threads), 0, 0, in, out, cyc, reps);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(n_wg), dim3(threads), 0, 0, in, out, cyc, reps);
                else hipLaunchKernelGGL(k<2>, dim3(n_wg), dim3(threads), 0, 0, in, out, cyc, reps);
            }
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> c(n_wg * threads / 64);
            CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            printf("%d waves per SIMD, %-30s %.2f cycles per MFMA per wave (median over %zu waves; min %.2f max %.2f)\n", threads / 256, names[mode],
                   c[c.size() / 2] / (16.0 * reps), c.size(), c.front() / (16.0 * reps), c.back() / (16.0 * reps));
        }
    return 0;
}
