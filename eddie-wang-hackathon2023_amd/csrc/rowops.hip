// Row-wise kernels: the consumers of the skinny GEMM's split-K slabs (bias, residual add,
// LayerNorm, GELU in one pass), token embedding, mel transpose, standalone LayerNorm.
//
// LayerNorm follows W/torch_model.py:25-27 (statistics and affine in fp32 on the fp16 row,
// eps 1e-5, result cast to fp16); kernel-level design reference generalLayerNorm
// (R/cpp/tensorrt_llm/kernels/layernormKernels.cu:62-188): one workgroup per row, two-pass
// (mean, then centred variance) on a register-resident row.
//
// All are HBM/L2-latency bound streaming kernels: 16-byte vector loads, 256 threads per row.
#include "common.h"
#include "kernels.h"

namespace wm {

constexpr int ROW_THREADS = 256;
constexpr int MAX_PER_THREAD = 24;     // rows up to 256 * 24 = 6144 columns (MLP hidden 5120)

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_THREADS / 64; ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(ROW_THREADS) void row_finish_kernel(RowFinishParams p) {
    __shared__ float red[ROW_THREADS / 64];
    const int m = blockIdx.x, tid = threadIdx.x;
    float v[MAX_PER_THREAD];
    const int per = (p.N + ROW_THREADS - 1) / ROW_THREADS;
    const size_t sstride = p.part_sstride ? (size_t)p.part_sstride : (size_t)p.M * p.ldp;
    // ---- gather: y = sum_s part + bias, fp16; then residual / gelu ----------------------------
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        if (i >= per) break;
        const int n = tid + i * ROW_THREADS;
        float x = 0.f;
        if (n < p.N) {
            if (p.mode != 2) {
                float y = 0.f;
                for (int s = 0; s < p.ksplit; ++s) y += p.part[(size_t)s * sstride + (size_t)m * p.ldp + n];
                if (p.bias) y += (float)p.bias[n];
                y = r16(y);
                if (p.mode == 1) {
                    x = r16(p.gelu_kind == 2 ? gelu_tanh(y) : gelu_erf(y));
                } else {
                    x = r16((float)p.x[(size_t)m * p.ldx + n] + y);
                }
            } else {
                x = (float)p.x[(size_t)m * p.ldx + n];
            }
        }
        v[i] = x;
    }
    if (p.mode == 1) {
#pragma unroll
        for (int i = 0; i < MAX_PER_THREAD; ++i) {
            if (i >= per) break;
            const int n = tid + i * ROW_THREADS;
            if (n < p.N) p.out[(size_t)m * p.ldo + n] = (h16)v[i];
        }
        return;
    }
    if (p.mode == 0 || p.mode == 3) {
#pragma unroll
        for (int i = 0; i < MAX_PER_THREAD; ++i) {
            if (i >= per) break;
            const int n = tid + i * ROW_THREADS;
            if (n < p.N) p.x[(size_t)m * p.ldx + n] = (h16)v[i];
        }
        if (p.mode == 3) return;
    }
    // ---- LayerNorm -------------------------------------------------------------------------
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        if (i >= per) break;
        if (tid + i * ROW_THREADS < p.N) s += v[i];
    }
    const float mean = block_sum(s, red) / (float)p.N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        if (i >= per) break;
        if (tid + i * ROW_THREADS < p.N) { const float d = v[i] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(block_sum(q, red) / (float)p.N + 1e-5f);
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        if (i >= per) break;
        const int n = tid + i * ROW_THREADS;
        if (n < p.N)
            p.out[(size_t)m * p.ldo + n] = (h16)((v[i] - mean) * rstd * (float)p.ln_g[n] + (float)p.ln_b[n]);
    }
}

int launch_row_finish(const RowFinishParams& p, hipStream_t stream) {
    WM_REQUIRE(p.N <= ROW_THREADS * MAX_PER_THREAD, "row_finish: N=%d too wide", p.N);
    WM_REQUIRE(p.M > 0, "row_finish: empty M");
    hipLaunchKernelGGL(row_finish_kernel, dim3(p.M), dim3(ROW_THREADS), 0, stream, p);
    WM_LAUNCH_CHECK(stream, "row_finish");
    return 0;
}

int launch_layernorm(const h16* x, int ldx, int M, int N, const h16* g, const h16* b, h16* out, int ldo,
                     hipStream_t stream) {
    RowFinishParams p{};
    p.mode = 2; p.M = M; p.N = N; p.x = const_cast<h16*>(x); p.ldx = ldx; p.ln_g = g; p.ln_b = b;
    p.out = out; p.ldo = ldo;
    return launch_row_finish(p, stream);
}

// ---- token embedding + positional embedding (whisper/model.py:257-260) --------------------------
// The embedding matrix is stored once, in the logits GEMM's fp16 tile-linear layout; a row is
// gathered as C/8 16-byte pieces: piece (kt, g) of token t sits at
//   ((t / 16) * (C / 32) + kt) * 1024 + ((t & 15) + 16 * g) * 16 bytes.
__global__ __launch_bounds__(256) void embed_kernel(EmbedParams p) {
    const int m = blockIdx.x;
    int tok = p.tokens[(size_t)(m / p.L) * p.tokens_ld + m % p.L];
    if (tok < 0) tok = 0;
    if (tok >= p.n_vocab) tok = p.n_vocab - 1;
    const int kt_total = p.C / 32;
    const unsigned char* base = (const unsigned char*)p.emb_tiles + (size_t)(tok >> 4) * kt_total * 1024 + (tok & 15) * 16;
    const h16* pos = p.pos + (size_t)(m % p.L) * p.C;
    for (int piece = threadIdx.x; piece < p.C / 8; piece += blockDim.x) {
        const int kt = piece >> 2, g = piece & 3;
        const half8v e = *(const half8v*)(base + (size_t)kt * 1024 + g * 256);
        const half8v q = *(const half8v*)(pos + piece * 8);
        half8v o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (h16)((float)e[j] + (float)q[j]);
        *(half8v*)(p.x + (size_t)m * p.ldx + piece * 8) = o;
    }
}

int launch_embed(const EmbedParams& p, hipStream_t stream) {
    WM_REQUIRE(p.C % 32 == 0, "embed: C=%d must be a multiple of 32", p.C);
    hipLaunchKernelGGL(embed_kernel, dim3(p.M), dim3(256), 0, stream, p);
    WM_LAUNCH_CHECK(stream, "embed");
    return 0;
}

// ---- mel [B][n_mels][T] (channel-major, what run.py hands over) -> token-major [B][T+2][n_mels]
// with one zero row before and after each utterance, so that conv1 (k3, s1, p1) is a GEMM over a
// strided view of this buffer.
__global__ __launch_bounds__(256) void mel_transpose_pad_kernel(const h16* mel, int n_mels, int T, h16* out) {
    __shared__ h16 tile[64][65];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const h16* src = mel + (size_t)b * n_mels * T;
    h16* dst = out + (size_t)b * (T + 2) * n_mels;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, t = i & 63;
        tile[c][t] = (c0 + c < n_mels && t0 + t < T) ? src[(size_t)(c0 + c) * T + t0 + t] : (h16)0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        if (c0 + c < n_mels && t0 + t < T) dst[(size_t)(t0 + t + 1) * n_mels + c0 + c] = tile[c][t];
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int c = threadIdx.x; c < n_mels; c += 256) {
            dst[c] = (h16)0.f;
            dst[(size_t)(T + 1) * n_mels + c] = (h16)0.f;
        }
    }
}

int launch_mel_transpose_pad(const h16* mel, int B, int n_mels, int T, h16* out, hipStream_t stream) {
    dim3 grid((T + 63) / 64, (n_mels + 63) / 64, B);
    hipLaunchKernelGGL(mel_transpose_pad_kernel, grid, dim3(256), 0, stream, mel, n_mels, T, out);
    WM_LAUNCH_CHECK(stream, "mel_transpose_pad");
    return 0;
}

__global__ void zero_pad_rows_kernel(h16* buf, int Tpad, int C) {
    h16* base = buf + (size_t)blockIdx.x * Tpad * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        base[c] = (h16)0.f;
        base[(size_t)(Tpad - 1) * C + c] = (h16)0.f;
    }
}

int launch_zero_pad_rows(h16* buf, int B, int Tpad, int C, hipStream_t stream) {
    hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(B), dim3(256), 0, stream, buf, Tpad, C);
    WM_LAUNCH_CHECK(stream, "zero_pad_rows");
    return 0;
}

// ---- int8 quantisation: sat_s8(rne(x * inv_scale))  (attention.py:340-348; QuantizeTensor plugin) ----
__global__ void quantize_i8_kernel(const h16* x, int8_t* q, long n, float inv_scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = rintf((float)x[i] * inv_scale);
        q[i] = (int8_t)fminf(127.f, fmaxf(-128.f, v));
    }
}

int launch_quantize_i8(const h16* x, int8_t* q, long n, float inv_scale, hipStream_t stream) {
    const int grid = (int)min((long)2048, (n + 255) / 256);
    hipLaunchKernelGGL(quantize_i8_kernel, dim3(max(grid, 1)), dim3(256), 0, stream, x, q, n, inv_scale);
    WM_LAUNCH_CHECK(stream, "quantize_i8");
    return 0;
}

}  // namespace wm
