// Lab check (not product code): wave_sum / wave_max of csrc/common.h (permlane swaps + DPP) against the __shfl_xor butterfly
// they replace, bit for bit, on random data.   hipcc -O3 --offload-arch=gfx950 -I include scripts/lab/wave_reduce_test.hip -o build/wave_reduce_test
#include <stdio.h>
#include <stdlib.h>
#include "../../eddie-wang-hackathon2023_amd/csrc/common.h"
using namespace wm;
__device__ float ref_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ float ref_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__global__ void k(const float* x, int n_waves, int* bad) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const float v = x[(size_t)w * 64 + (threadIdx.x & 63)];
    const float a = wave_sum(v), b = ref_sum(v), c = wave_max(v), d = ref_max(v);
    if (__float_as_uint(a) != __float_as_uint(b) || __float_as_uint(c) != __float_as_uint(d)) atomicAdd(bad, 1);
}
int main() {
    const int n_waves = 1 << 16;
    float* h = (float*)malloc((size_t)n_waves * 64 * 4);
    srand(7);
    for (size_t i = 0; i < (size_t)n_waves * 64; ++i) h[i] = ((rand() % 20001) - 10000) / 37.0f * ((rand() & 3) ? 1.0f : 1e-4f);
    float* d; int* bad; int hb = 0;
    hipMalloc(&d, (size_t)n_waves * 64 * 4); hipMalloc(&bad, 4);
    hipMemcpy(d, h, (size_t)n_waves * 64 * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
    k<<<n_waves / 4, 256>>>(d, n_waves, bad);
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("%d waves x 64 lanes: %d lanes differ from the __shfl_xor butterfly\n", n_waves, hb);
    return hb != 0;
}
