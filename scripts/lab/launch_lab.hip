// Lab: floor of dependent kernel chains on this GPU (eager and hipGraph), for tiny kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
__global__ void k_chain(const float* a, float* b, int n) {   // read a (written by previous kernel), write b
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i] + 1.0f; }
int main() {
    float *a, *b; const int n = 64 * 1280;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int N = 2000;
    auto timeit = [&](const char* name, auto f) {
        for (int i = 0; i < 50; ++i) f(i);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < N; ++i) f(i);
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        printf("%-44s %.2f us per kernel\n", name, us / N);
    };
    timeit("eager empty <<<1,64>>>", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
    timeit("eager empty <<<256,256>>>", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); });
    timeit("eager touch 320 WG x 256 (80K floats)", [&](int) { hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, s, a, n); });
    timeit("eager chain a->b->a 320 WG", [&](int i) { if (i & 1) hipLaunchKernelGGL(k_chain, dim3(n / 256), dim3(256), 0, s, b, a, n); else hipLaunchKernelGGL(k_chain, dim3(n / 256), dim3(256), 0, s, a, b, n); });
    // graph of 400 chained kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 400; ++i) { if (i & 1) hipLaunchKernelGGL(k_chain, dim3(n / 256), dim3(256), 0, s, b, a, n); else hipLaunchKernelGGL(k_chain, dim3(n / 256), dim3(256), 0, s, a, b, n); }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    auto t1 = std::chrono::high_resolution_clock::now();
    hipStreamSynchronize(s);
    auto t2 = std::chrono::high_resolution_clock::now();
    printf("graph of 400 chained kernels: host %.1f us per launch, %.2f us per kernel wall\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 10, std::chrono::duration<double, std::micro>(t2 - t0).count() / 4000);
    // two streams, two graphs concurrently
    hipStream_t s2; CK(hipStreamCreate(&s2));
    float *c, *d; CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&d, n * 4));
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(s2, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 400; ++i) { if (i & 1) hipLaunchKernelGGL(k_chain, dim3(n / 256), dim3(256), 0, s2, d, c, n); else hipLaunchKernelGGL(k_chain, dim3(n / 256), dim3(256), 0, s2, c, d, n); }
    CK(hipStreamEndCapture(s2, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    hipGraphLaunch(ge, s); hipGraphLaunch(ge2, s2); hipDeviceSynchronize();
    t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < 10; ++r) { hipGraphLaunch(ge, s); hipGraphLaunch(ge2, s2); }
    hipDeviceSynchronize();
    t2 = std::chrono::high_resolution_clock::now();
    printf("two graphs on two streams: %.2f us per kernel-pair wall (8000 kernels total)\n", std::chrono::duration<double, std::micro>(t2 - t0).count() / 4000);
    return 0;
}
