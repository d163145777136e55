// Lab harness (not product code): the K64-slot variant (scripts/lab/gemm_f16r.hip: full-line LDS-DMA pieces) against the product kernel
// (csrc/gemm_f16p.hip) on the encoder's shapes: results compared element by element, then timed in interleaved rounds in ONE process, random data.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include scripts/lab/gemm_lab5.hip -o /tmp/gemm_lab5 && /tmp/gemm_lab5 [M]
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../eddie-wang-hackathon2023_amd/csrc/gemm_f16.hip"
#include "../../eddie-wang-hackathon2023_amd/csrc/gemm_f16p.hip"
#include "gemm_f16r.hip"
namespace wm {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int lab_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
int post_launch_check(hipStream_t, const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 2; } return 0; }
}
using namespace wm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void fill(h16* p, size_t n, float scale, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (h16)(((int)(x & 0xffff) - 32768) / 32768.0f * scale);
    }
}
__global__ void diff(const h16* a, const h16* b, size_t n, float* out) {
    float m = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf((float)a[i] - (float)b[i]));
    atomicMax((unsigned*)out, __float_as_uint(m));
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 192000;
    const size_t maxMN = (size_t)M * 5120;
    h16 *A, *W, *C0, *C1, *bias, *R;
    CK(hipMalloc(&A, maxMN * 2)); CK(hipMalloc(&W, (size_t)5120 * 5120 * 2)); CK(hipMalloc(&C0, maxMN * 2)); CK(hipMalloc(&C1, maxMN * 2));
    CK(hipMalloc(&bias, 5120 * 2)); CK(hipMalloc(&R, maxMN * 2));
    fill<<<2048, 256>>>(A, maxMN, 1.0f, 1); fill<<<2048, 256>>>(W, (size_t)5120 * 5120, 0.03f, 2); fill<<<64, 256>>>(bias, 5120, 0.2f, 3); fill<<<2048, 256>>>(R, maxMN, 1.0f, 4);
    float* dmax; CK(hipMalloc(&dmax, 4));
    struct S { const char* name; int N, K, act; bool res; int colscale_n; } shapes[] = {
        {"qkv   N=3840 K=1280 colscale", 3840, 1280, 0, false, 2560}, {"out   N=1280 K=1280 residual", 1280, 1280, 0, true, 0},
        {"mlp1  N=5120 K=1280 gelu", 5120, 1280, 1, false, 0}, {"mlp2  N=1280 K=5120 residual", 1280, 5120, 0, true, 0},
        {"plain N=4096 K=4096 (M as given)", 4096, 4096, 0, false, 0}, {"plain N=1280 K=5120", 1280, 5120, 0, false, 0}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto sh : shapes) {
        GemmBigParams p{};
        p.A = A; p.lda = sh.K; p.M = M; p.K = sh.K; p.W = W; p.N = sh.N; p.bias = bias; p.ldc = sh.N; p.act = sh.act;
        if (sh.res) { p.residual = R; p.ldr = sh.N; }
        if (sh.colscale_n) { p.colscale_n = sh.colscale_n; p.colscale = 0.35355339f; }
        // correctness: the M tail too (M - 100 rows)
        GemmBigParams q = p; q.M = M - 100;
        CK(hipMemset(C0, 0, maxMN * 2)); CK(hipMemset(C1, 0, maxMN * 2));
        q.C = C0; launch_gemm_f16p(q, 0);
        q.C = C1; launch_gemm_f16r(q, 0);
        CK(hipMemset(dmax, 0, 4));
        diff<<<2048, 256>>>(C0, C1, (size_t)M * sh.N, dmax);
        float hm; CK(hipMemcpy(&hm, dmax, 4, hipMemcpyDeviceToHost));
        printf("%s: max |f16p - f16r| = %g\n", sh.name, hm);
        // timing: interleaved rounds
        double best[2] = {1e9, 1e9}, sum[2] = {0, 0};
        const int rounds = 5;
        for (int r = 0; r < rounds + 1; ++r)
            for (int v = 0; v < 2; ++v) {
                p.C = v ? C1 : C0;
                CK(hipEventRecord(e0));
                for (int it = 0; it < 3; ++it) { if (v) launch_gemm_f16r(p, 0); else launch_gemm_f16p(p, 0); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
                if (r > 0) { best[v] = std::min(best[v], (double)ms); sum[v] += ms; }
            }
        const double fl = 2.0 * M * sh.N * sh.K;
        printf("    f16p: %.3f ms avg (%.0f TF/s), best %.0f TF/s | f16r: %.3f ms avg (%.0f TF/s), best %.0f TF/s\n", sum[0] / rounds, fl / (sum[0] / rounds) * 1e-9,
               fl / best[0] * 1e-9, sum[1] / rounds, fl / (sum[1] / rounds) * 1e-9, fl / best[1] * 1e-9);
    }
    return 0;
}
