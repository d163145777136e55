// Round 4 lab: what does ONE all-to-all edge cost inside a persistent launch on this chip?
//
// The question behind a persistent per-layer decode kernel for small batches (task 2): a decoder layer at batch 1
// is nine launches of ~5.5 us; inside one launch the stages would be separated by all-to-all hand-offs of an activation
// vector (1280 .. 5120 fp16 values x rows) from the CUs that produced its slices to every CU.  This harness times that
// hand-off in the form MI355X_MICROARCH.md prices cheapest (its rows `allgather` / R2 of Guideline 16): 8-byte {tag, value}
// granules written with ONE agent-scope (sc1, write-through) store each, consumers sweeping ALL granules with sc1 loads
// until every tag carries the round's epoch -- no flag, no fence, no barrier -- against (b) a kernel boundary per round
// (a captured graph of trivial dependent kernels doing the same exchange through plain memory).
//
//   every round: WG w publishes its slice (G granules per row) computed from what it gathered in the previous round;
//   then every WG gathers all NWG * G * ROWS granules into LDS; rounds are dependent.  Checked against a host simulation.
//
//   hipcc -O3 --offload-arch=gfx950 scripts/lab/edge_lab.hip -o build/lab/edge_lab && build/lab/edge_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int NWG = 256;

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) { return (a * 2654435761u) ^ (b + 0x9e3779b9u + (a << 6) + (a >> 2)); }

// ROWS activation rows, G granules per WG and row (vector width = 2 * G * NWG halves), SWEEPERS waves share the sweep
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
// 16-byte sc1 load (two granules) -- inline asm: the builtin atomic load is 8 bytes wide at most
__device__ __forceinline__ u32x4v load16_sc1(const unsigned long long* p) {
    u32x4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// WIDE = 1: the sweep reads two granules per 16-byte load, all loads of a pass in flight together; SLEEP: s_sleep argument between passes
template <int ROWS, int G, int SWEEPERS, int SLEEP>
__global__ __launch_bounds__(256) void edge16_kernel(unsigned long long* gran, int rounds, unsigned* out, unsigned* tmo) {
    __shared__ unsigned vec[ROWS * NWG * G];
    const int w = blockIdx.x, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    constexpr int N = NWG * G;
    static_assert((ROWS * N) % (128 * SWEEPERS) == 0, "whole 16-byte loads per lane");
    for (int i = tid; i < ROWS * N; i += 256) vec[i] = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const unsigned epoch = r + 1;
        unsigned long long* buf = gran + (size_t)(r & 1) * ROWS * N;
        if (tid < ROWS * G) {
            const int row = tid / G, g = tid % G;
            unsigned v = mix(vec[row * N + ((w * G + g) * 7 + r) % N], w * 131 + g);
            v = mix(v, vec[row * N + (w + 17 * g + 3 * r) % N]);
            __hip_atomic_store((gu64*)(buf + row * N + w * G + g), ((unsigned long long)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wid < SWEEPERS) {
            constexpr int PER_WAVE = ROWS * N / SWEEPERS;        // granules
            constexpr int LOADS = PER_WAVE / 128;                // 16-byte loads per lane
            const int base = wid * PER_WAVE;
            unsigned spins = 0;
            for (;;) {
                u32x4v val[LOADS];
                // all loads of the pass in flight, one wait
#pragma unroll
                for (int k = 0; k < LOADS; ++k)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(val[k]) : "v"(buf + base + (k * 64 + lane) * 2) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bool ok = true;
#pragma unroll
                for (int k = 0; k < LOADS; ++k) { asm volatile("" : "+v"(val[k])); ok &= val[k][1] == epoch && val[k][3] == epoch; }
                if (__all(ok)) {
#pragma unroll
                    for (int k = 0; k < LOADS; ++k) { vec[base + (k * 64 + lane) * 2] = val[k][0]; vec[base + (k * 64 + lane) * 2 + 1] = val[k][2]; }
                    break;
                }
                if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(tmo, 1u); break; }
                __builtin_amdgcn_s_sleep(SLEEP);
            }
        }
        __syncthreads();
    }
    if (tid == 0) { unsigned h = 0; for (int i = 0; i < ROWS * N; ++i) h = mix(h, vec[i]); out[w] = h; }
}

template <int ROWS, int G, int SWEEPERS>
__global__ __launch_bounds__(256) void edge_kernel(unsigned long long* gran /* [2][ROWS][NWG * G] */, int rounds, unsigned* out, unsigned* tmo) {
    __shared__ unsigned vec[ROWS * NWG * G];
    const int w = blockIdx.x, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    constexpr int N = NWG * G;                       // granules per row
    for (int i = tid; i < ROWS * N; i += 256) vec[i] = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const unsigned epoch = r + 1;
        unsigned long long* buf = gran + (size_t)(r & 1) * ROWS * N;
        // ---- "compute" this WG's slice from the gathered vector of the previous round, publish it (ONE sc1 store per granule)
        if (tid < ROWS * G) {
            const int row = tid / G, g = tid % G;
            unsigned v = mix(vec[row * N + ((w * G + g) * 7 + r) % N], w * 131 + g);
            v = mix(v, vec[row * N + (w + 17 * g + 3 * r) % N]);
            __hip_atomic_store((gu64*)(buf + row * N + w * G + g), ((unsigned long long)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- gather: SWEEPERS waves sweep their share of the granules until every tag matches
        if (wid < SWEEPERS) {
            constexpr int PER_WAVE = (ROWS * N + SWEEPERS - 1) / SWEEPERS;
            constexpr int LOADS = (PER_WAVE + 63) / 64;
            const int base = wid * PER_WAVE;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                unsigned val[LOADS];
#pragma unroll
                for (int k = 0; k < LOADS; ++k) {
                    const int i = base + k * 64 + lane;
                    if (i < base + PER_WAVE && i < ROWS * N) {
                        const unsigned long long x = __hip_atomic_load((gu64*)(buf + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        val[k] = (unsigned)x;
                        ok &= (unsigned)(x >> 32) == epoch;
                    } else val[k] = 0;
                }
                if (__all(ok)) {
#pragma unroll
                    for (int k = 0; k < LOADS; ++k) {
                        const int i = base + k * 64 + lane;
                        if (i < base + PER_WAVE && i < ROWS * N) vec[i] = val[k];
                    }
                    break;
                }
                if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(tmo, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (tid == 0) { unsigned h = 0; for (int i = 0; i < ROWS * N; ++i) h = mix(h, vec[i]); out[w] = h; }
}

// (b) the same exchange with a kernel boundary per round: plain stores, plain loads
template <int ROWS, int G>
__global__ __launch_bounds__(256) void round_kernel(const unsigned* prev /* [ROWS][N] */, unsigned* next, int r) {
    constexpr int N = NWG * G;
    const int w = blockIdx.x, tid = threadIdx.x;
    __shared__ unsigned vec[ROWS * N];
    for (int i = tid; i < ROWS * N; i += 256) vec[i] = prev[i];
    __syncthreads();
    if (tid < ROWS * G) {
        const int row = tid / G, g = tid % G;
        unsigned v = mix(vec[row * N + ((w * G + g) * 7 + r) % N], w * 131 + g);
        v = mix(v, vec[row * N + (w + 17 * g + 3 * r) % N]);
        next[row * N + w * G + g] = v;
    }
}

template <int ROWS, int G>
static unsigned host_hash(int rounds) {
    constexpr int N = NWG * G;
    std::vector<unsigned> vec(ROWS * N, 0), nxt(ROWS * N, 0);
    auto mixh = [](unsigned a, unsigned b) { return (a * 2654435761u) ^ (b + 0x9e3779b9u + (a << 6) + (a >> 2)); };
    for (int r = 0; r < rounds; ++r) {
        for (int w = 0; w < NWG; ++w)
            for (int row = 0; row < ROWS; ++row)
                for (int g = 0; g < G; ++g) {
                    unsigned v = mixh(vec[row * N + ((w * G + g) * 7 + r) % N], w * 131 + g);
                    v = mixh(v, vec[row * N + (w + 17 * g + 3 * r) % N]);
                    nxt[row * N + w * G + g] = v;
                }
        vec.swap(nxt);
    }
    unsigned h = 0;
    for (int i = 0; i < ROWS * N; ++i) h = mixh(h, vec[i]);
    return h;
}

template <int ROWS, int G, int SWEEPERS>
static void run(int rounds) {
    constexpr int N = NWG * G;
    unsigned long long* gran; unsigned *out, *tmo;
    CHECK(hipMalloc(&gran, sizeof(unsigned long long) * 2 * ROWS * N));
    CHECK(hipMalloc(&out, sizeof(unsigned) * NWG));
    CHECK(hipMalloc(&tmo, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    std::vector<unsigned> h(NWG);
    unsigned tm = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(gran, 0, sizeof(unsigned long long) * 2 * ROWS * N));
        CHECK(hipMemset(tmo, 0, 16));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((edge_kernel<ROWS, G, SWEEPERS>), dim3(NWG), dim3(256), 0, 0, gran, rounds, out, tmo);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(h.data(), out, sizeof(unsigned) * NWG, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&tm, tmo, 4, hipMemcpyDeviceToHost));
    }
    const unsigned want = host_hash<ROWS, G>(rounds);
    int bad = 0;
    for (int w = 0; w < NWG; ++w) bad += h[w] != want;
    // (b) one kernel per round, replayed from a graph
    unsigned *a, *b;
    CHECK(hipMalloc(&a, sizeof(unsigned) * ROWS * N)); CHECK(hipMalloc(&b, sizeof(unsigned) * ROWS * N));
    CHECK(hipMemset(a, 0, sizeof(unsigned) * ROWS * N));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipGraph_t graph; hipGraphExec_t exec;
    const int gr = 200;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int r = 0; r < gr; ++r) hipLaunchKernelGGL((round_kernel<ROWS, G>), dim3(NWG), dim3(256), 0, s, (r & 1) ? b : a, (r & 1) ? a : b, r);
    CHECK(hipStreamEndCapture(s, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    float gbest = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, s));
        CHECK(hipGraphLaunch(exec, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < gbest) gbest = ms;
    }
    printf("rows %2d, %4d halves wide (%d granules/WG/row), %d sweeping wave(s): in-launch edge %.2f us per round (%s, timeouts %u) | kernel boundary per round %.2f us\n",
           ROWS, 2 * N, G, SWEEPERS, best * 1e3f / rounds, bad ? "WRONG" : "checked", tm, gbest * 1e3f / gr);
    CHECK(hipFree(gran)); CHECK(hipFree(out)); CHECK(hipFree(tmo)); CHECK(hipFree(a)); CHECK(hipFree(b));
}

template <int ROWS, int G, int SWEEPERS, int SLEEP>
static void run16(int rounds) {
    constexpr int N = NWG * G;
    unsigned long long* gran; unsigned *out, *tmo;
    CHECK(hipMalloc(&gran, sizeof(unsigned long long) * 2 * ROWS * N));
    CHECK(hipMalloc(&out, sizeof(unsigned) * NWG));
    CHECK(hipMalloc(&tmo, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    std::vector<unsigned> h(NWG);
    unsigned tm = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(gran, 0, sizeof(unsigned long long) * 2 * ROWS * N));
        CHECK(hipMemset(tmo, 0, 16));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((edge16_kernel<ROWS, G, SWEEPERS, SLEEP>), dim3(NWG), dim3(256), 0, 0, gran, rounds, out, tmo);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(h.data(), out, sizeof(unsigned) * NWG, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&tm, tmo, 4, hipMemcpyDeviceToHost));
    }
    const unsigned want = host_hash<ROWS, G>(rounds);
    int bad = 0;
    for (int w = 0; w < NWG; ++w) bad += h[w] != want;
    printf("rows %2d, %4d halves wide, %d sweeping wave(s), 16-byte loads, s_sleep %d: in-launch edge %.2f us per round (%s, timeouts %u)\n",
           ROWS, 2 * N, SWEEPERS, SLEEP, best * 1e3f / rounds, bad ? "WRONG" : "checked", tm);
    CHECK(hipFree(gran)); CHECK(hipFree(out)); CHECK(hipFree(tmo));
}

int main() {
    const int rounds = 2000;
    run<1, 2, 1>(rounds);      // 1024 halves
    run<1, 4, 1>(rounds);      // 2048 halves
    run<1, 4, 2>(rounds);
    run<1, 10, 2>(rounds);     // 5120 halves
    run<1, 10, 4>(rounds);
    run<4, 4, 4>(rounds);
    run<8, 4, 4>(rounds);
    run<16, 4, 4>(rounds);
    run16<1, 4, 1, 1>(rounds);     // 2048 halves, one wave, 8 loads of 16 bytes per lane
    run16<1, 4, 1, 8>(rounds);
    run16<1, 4, 2, 8>(rounds);
    run16<1, 10, 1, 4>(rounds);    // 5120 halves, one wave: 20 loads per lane
    run16<1, 10, 2, 4>(rounds);
    run16<1, 10, 4, 4>(rounds);
    run16<2, 4, 1, 4>(rounds);
    run16<4, 4, 2, 4>(rounds);
    return 0;
}
