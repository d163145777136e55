// Lab harness (not product code): the ns1 experiment's throughput half -- the persistent GEMM of csrc/gemm_f16p.hip with int8
// operands on v_mfma_i32_16x16x64_i8 (same tile, same LDS ring, same alternating wave groups; a 64-byte LDS row now holds 64
// int8 inputs, so a stage is twice as deep in K for the same bytes and the same MFMA cycles) next to the fp16 kernel on the
// encoder's shapes, plus the per-token quantisation pass int8 activations would need.  Accuracy half: scripts/experiments/
// ns1_w8a8_accuracy.py.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include scripts/lab/gemm_i8_lab.hip -o /tmp/gemm_i8_lab
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "../../eddie-wang-hackathon2023_amd/csrc/gemm_f16.hip"
#include "../../eddie-wang-hackathon2023_amd/csrc/gemm_f16p.hip"
namespace wm {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int post_launch_check(hipStream_t, const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 2; } return 0; }
typedef int int4v __attribute__((ext_vector_type(4)));
struct I8Params { const int8_t* A; const int8_t* W; const float* row_scale; const h16* col_scale; h16* C; int M, N, K; };

template <int STAGES>
__global__ __launch_bounds__(512) void gemm_i8_kernel(I8Params p) {
    using namespace f16p;      // same tile geometry: a stage is 256 + 256 rows of 64 BYTES (64 int8 inputs instead of 32 halves)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int WAIT = 4 * STAGES - 10;            // DMA pieces of this wave that may still be in flight when stage s + 1 must have landed
    constexpr int N_STORES = 32;                     // store instructions of one wave's epilogue (8 pieces x 4)

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;           // 4 (M) x 2 (N) waves, each 64 rows x 128 channels
    const int g = lane >> 4;

    // ---- this workgroup's tiles: XCD x owns the contiguous band [lo, hi) of the tile list (channel tile fastest) -----
    const int nt_n = p.N / BN, nt_m = (p.M + BM - 1) / BM, n_tiles = nt_n * nt_m;
    const int per_xcd = gridDim.x >> 3;              // gridDim.x is a multiple of 8
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3;
    const int band = (n_tiles + 7) >> 3;
    const int lo = min(n_tiles, xcd * band), hi = min(n_tiles, lo + band);
    const int my_tiles = (hi - lo - j0 + per_xcd - 1) > 0 ? (hi - lo - j0 + per_xcd - 1) / per_xcd : 0;
    if (my_tiles == 0) return;
    const int nk = p.K / 64;                         // stages per tile
    const int total_stages = my_tiles * nk;

    // ---- loader: per stage this wave requests 2 A pieces and 2 W pieces of 16 rows x 64 B, as two "halves" (one A and one
    // W piece each).  Addresses are a wave-uniform base per tile (advanced by 64 B per stage) plus a 32-bit lane offset ---
    const unsigned char* a_base = nullptr;           // row `row0` of the tile, at the stage's K offset
    const unsigned char* w_base = nullptr;
    uint32_t a_lane[2], w_lane[2];
    auto row_offset = [&](int gr) -> size_t {        // element offset of row gr of A (plain or a strided view)
        return (size_t)gr * p.K;
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wid + NWAVE * q) * 16 + (lane >> 2);                       // row inside the 256-row part
        const int c = (lane & 3) ^ ((4 - ((r >> 2) & 3)) & 3);                    // source chunk for this LDS slot
        w_lane[q] = (uint32_t)r * (uint32_t)p.K + c * 16;
    }
    auto set_tile = [&](int t) {                     // loader -> tile number t of this workgroup, K offset 0
        const int tile = lo + j0 + t * per_xcd;
        const int tm = tile / nt_n, tn = tile - tm * nt_n;
        const size_t off0 = row_offset(tm * BM);
        a_base = (const unsigned char*)p.A + off0;
        w_base = (const unsigned char*)p.W + (size_t)tn * BN * p.K;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wid + NWAVE * q) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((4 - ((r >> 2) & 3)) & 3);
            int gr = tm * BM + r;
            if (gr > p.M - 1) gr = p.M - 1;          // rows past the end re-read the last row (never stored)
            a_lane[q] = (uint32_t)(row_offset(gr) - off0) + c * 16;
        }
    };
    int load_ks = 0, load_tile = 0, issued = 0;      // stage whose pieces are requested next (issued = its stream number)
    auto issue_half = [&](int half) {                // this wave's A piece and W piece number `half` of stream stage `issued`
        if (issued >= total_stages) return;
        unsigned char* slot = smem + (issued % STAGES) * STAGE + (wid + NWAVE * half) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + a_lane[half]),
                                         (__attribute__((address_space(3))) void*)slot, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_base + w_lane[half]),
                                         (__attribute__((address_space(3))) void*)(slot + A_PART), 16, 0, 0);
        if (half == 1) {
            a_base += 64; w_base += 64;
            ++issued;
            if (++load_ks == nk) { load_ks = 0; ++load_tile; if (load_tile < my_tiles) set_tile(load_tile); }
        }
    };
    set_tile(0);
    // what the steady-state schedule below assumes was requested before the first stage is multiplied: stages 0 .. STAGES - 3
    // complete and the first half of stage STAGES - 2
#pragma unroll 1
    for (int s2 = 0; s2 < 2 * (STAGES - 2) + 1; ++s2) issue_half(s2 & 1);

    // ---- fragment addresses inside a stage --------------------------------------------------------------------------
    const int sw = (4 - ((lane >> 2) & 3)) & 3;                                   // the rows a lane reads have (r >> 2) & 3 = (lane >> 2) & 3
    const int a_off = (wr * 64 + (lane & 15)) * 64 + ((g ^ sw) << 4);
    const int b_off = A_PART + (wc * 128 + (lane & 15)) * 64 + ((g ^ sw) << 4);

    int4v acc[4][8];
    int4v af[4], bx[4];                             // A rows (4 blocks); W channels, first or second 64 of this wave (4 blocks)

    {   // per-channel scales (fp16 weight scales) behind the ring, as the fp16 kernel keeps its bias there
        h16* cs0 = (h16*)(smem + STAGES * STAGE);
        for (int c = tid; c < p.N; c += 512) cs0[c] = p.col_scale[c];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // The two waves of a SIMD (w and w + 4) alternate: while one multiplies (16 MFMAs, nothing else), the other reads its next
    // fragments from LDS, requests DMA pieces and waits -- the matrix pipe always has one wave feeding it.  The workgroup's
    // barriers are the clock of that alternation; waves 4-7 run one barrier behind waves 0-3.
    if (wid >= 4) __builtin_amdgcn_s_barrier();

    int cons = 0;                                    // stream stage being multiplied
    int ks = 0, t = 0;
    bool after_epilogue = false;                     // the next stage wait has this wave's epilogue stores in its queue
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = int4v{0, 0, 0, 0};
    };
    zero_acc();
    for (;;) {
        const unsigned char* st = smem + (cons % STAGES) * STAGE;
        // ---- first half of the channels: fragments, DMA requests | barrier | 16 MFMAs | barrier ------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const int4v*)(st + a_off + i * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) bx[j] = *(const int4v*)(st + b_off + j * 1024);
        issue_half(1);                               // completes stage cons + STAGES - 2
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bx[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- second half ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) bx[j] = *(const int4v*)(st + b_off + (4 + j) * 1024);
        issue_half(0);                               // opens stage cons + STAGES - 1
        // Stage cons + 1 is read by waves 0-3 two barriers from here (one for waves 4-7): this wave's pieces of it must have
        // landed before the next barrier.  Its last pieces were requested STAGES - 3 stages ago; WAIT younger requests may stay
        // in flight -- plus, on the first stage after an epilogue, the epilogue's stores, which sit behind those pieces in the
        // queue and must not be waited for here (they drain while the next stage is multiplied).
        if (issued >= total_stages) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAIT + N_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WAIT) : "memory");
        after_epilogue = false;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][4 + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bx[j], af[i], acc[i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        ++cons;
        if (++ks < nk) continue;

        // ================================ epilogue: y = fp16(acc * s_token * s_channel) ====================================
        const int tile = lo + j0 + t * per_xcd;
        const int tm = tile / nt_n, tn = tile - tm * nt_n;
        const int row0 = tm * BM, col0 = tn * BN;
        int le = lane;
        asm volatile("" : "+v"(le));
        const int ge = le >> 4, rl = le & 15;
        const int colw = col0 + wc * 128 + ge * 4;
        const h16* cs = (const h16*)(smem + STAGES * STAGE) + colw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + wr * 64 + i * 16 + rl;
            const float rs = p.row_scale[row < p.M ? row : p.M - 1];
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                if (row < p.M) {
                    h16* crow = p.C + (size_t)row * p.N + colw + jh * 64;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const half4v c4 = *(const half4v*)(cs + jh * 64 + j * 16);
                        half4v o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (h16)((float)acc[i][jh * 4 + j][r] * rs * (float)c4[r]);
                        *(half4v*)(crow + j * 16) = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the store count the next stage wait adds is exact only for a tile without an M tail (rows past M skip their stores)
        if (row0 + BM <= p.M) after_epilogue = true; else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ks = 0;
        if (++t == my_tiles) break;
        zero_acc();
    }
    if (wid < 4) __builtin_amdgcn_s_barrier();       // waves 4-7 ran one barrier behind
}


// per-token dynamic quantisation: s_t = max|x_t| / 127, q = clip(rne(x / s_t)) -- one wave per row
__global__ __launch_bounds__(256) void quantize_rows_kernel(const h16* x, int M, int K, int8_t* q, float* scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const h16* xr = x + (size_t)row * K;
    float mx = 0.f;
    for (int c = lane * 8; c < K; c += 512) { const half8v v = *(const half8v*)(xr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf((float)v[e])); }
    mx = wave_max(mx);
    const float s = fmaxf(mx, 1e-8f) / 127.f, inv = 1.f / s;
    for (int c = lane * 8; c < K; c += 512) { const half8v v = *(const half8v*)(xr + c);
        union { int8_t b[8]; uint2 u; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.b[e] = (int8_t)fminf(127.f, fmaxf(-127.f, rintf((float)v[e] * inv)));
        *(uint2*)(q + (size_t)row * K + c) = o.u; }
    if (lane == 0) scale[row] = s;
}
}  // namespace wm
using namespace wm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void fill16(h16* p, size_t n, float scale, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (h16)(((int)(x & 0xffff) - 32768) / 32768.0f * scale); }
}
__global__ void fill8(int8_t* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13; p[i] = (int8_t)(x & 0xff); }
}
__global__ void ref_check(const int8_t* A, const int8_t* W, const float* rs, const h16* cs, const h16* C, int M, int N, int K, float* maxerr) {
    const int row = blockIdx.x * 97 % M, col = (blockIdx.x * 31 + threadIdx.x) % N;          // a sample of outputs, exact integer dot products
    long acc = 0;
    for (int k = 0; k < K; ++k) acc += (int)A[(size_t)row * K + k] * (int)W[(size_t)col * K + k];
    const float want = (float)(h16)((float)(int)acc * rs[row] * (float)cs[col]);
    atomicMax((unsigned*)maxerr, __float_as_uint(fabsf(want - (float)C[(size_t)row * N + col])));
}
int main(int argc, char** argv) {
    setenv("WM_GEMM_ROUND1", "0", 1);
    const int M = argc > 1 ? atoi(argv[1]) : 192000;
    const size_t maxMK = (size_t)M * 5120;
    h16 *A16, *W16, *C, *bias, *cs; int8_t *A8, *W8; float *rs, *err;
    CK(hipMalloc(&A16, maxMK * 2)); CK(hipMalloc(&W16, (size_t)5120 * 5120 * 2)); CK(hipMalloc(&C, maxMK * 2)); CK(hipMalloc(&bias, 5120 * 2));
    CK(hipMalloc(&A8, maxMK)); CK(hipMalloc(&W8, (size_t)5120 * 5120)); CK(hipMalloc(&rs, (size_t)M * 4)); CK(hipMalloc(&cs, 5120 * 2)); CK(hipMalloc(&err, 4));
    fill16<<<2048, 256>>>(A16, maxMK, 1.0f, 1); fill16<<<2048, 256>>>(W16, (size_t)5120 * 5120, 0.03f, 2); fill16<<<64, 256>>>(bias, 5120, 0.2f, 3);
    fill8<<<2048, 256>>>(W8, (size_t)5120 * 5120, 5); fill16<<<64, 256>>>(cs, 5120, 0.01f, 6);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)gemm_i8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * f16p::STAGE + f16p::MAX_N * 2));
    struct S { const char* name; int N, K; } shapes[] = {{"qkv  N=3840 K=1280", 3840, 1280}, {"out  N=1280 K=1280", 1280, 1280}, {"mlp1 N=5120 K=1280", 5120, 1280}, {"mlp2 N=1280 K=5120", 1280, 5120}};
    for (auto sh : shapes) {
        const int N = sh.N, K = sh.K;
        auto tq = [&]() { quantize_rows_kernel<<<(M + 3) / 4, 256>>>(A16, M, K, A8, rs); };
        I8Params q{A8, W8, rs, cs, C, M, N, K};
        auto t8 = [&]() { hipLaunchKernelGGL(gemm_i8_kernel<4>, dim3(256), dim3(512), 4 * f16p::STAGE + f16p::MAX_N * 2, 0, q); };
        GemmBigParams p{}; p.A = A16; p.lda = K; p.M = M; p.K = K; p.W = W16; p.N = N; p.bias = bias; p.C = C; p.ldc = N;
        auto t16 = [&]() { launch_gemm_f16p(p, 0); };
        tq(); t8(); CK(hipMemset(err, 0, 4));
        ref_check<<<512, 64>>>(A8, W8, rs, cs, C, M, N, K, err);
        float herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        double ms[3] = {0, 0, 0};
        for (int r = 0; r < 6; ++r)
            for (int v = 0; v < 3; ++v) {
                CK(hipEventRecord(e0));
                for (int it = 0; it < 3; ++it) { if (v == 0) t16(); else if (v == 1) t8(); else tq(); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r > 0) ms[v] += t / 3 / 5;
            }
        const double fl = 2.0 * M * N * K;
        printf("%s M=%d: fp16 MFMA %.3f ms (%.0f TFLOP/s) | int8 MFMA %.3f ms (%.0f TOP/s), sampled max |err| vs exact = %g | per-token quantisation of the input %.3f ms -> int8 path %.3f ms = %.2fx\n",
               sh.name, M, ms[0], fl / ms[0] * 1e-9, ms[1], fl / ms[1] * 1e-9, herr, ms[2], ms[1] + ms[2], ms[0] / (ms[1] + ms[2]));
    }
    return 0;
}
