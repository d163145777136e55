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

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum_nomfma(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

constexpr int ROW_MAX_N = 6144;        // MLP hidden of large-v2 is 5120

__device__ __forceinline__ float4 ld_h4(const h16* p) {
    const half4v v = *(const half4v*)p;
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st_h4(h16* p, float4 v) {
    half4v o = {(h16)v.x, (h16)v.y, (h16)v.z, (h16)v.w};
    *(half4v*)p = o;
}

// One workgroup per row; the row lives in LDS as fp32 between the passes (no per-thread arrays:
// a runtime-indexed register array would be demoted to scratch).  4 elements per thread per trip,
// 16-byte loads of the fp32 slabs, 8-byte loads/stores of fp16 data.
__global__ __launch_bounds__(ROW_THREADS) void row_finish_kernel(RowFinishParams p) {
    // issue priority over the other groups' K/V stream waves (see gemm_skinny_kernel)
    __builtin_amdgcn_s_setprio(3);

    __shared__ float red[ROW_THREADS / 64];
    __shared__ __attribute__((aligned(16))) float row[ROW_MAX_N];
    const int m = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const size_t sstride = p.part_sstride ? (size_t)p.part_sstride : (size_t)p.M * p.ldp;
    const int n4 = p.N >> 2;
    float sum = 0.f;
    // mode 1 (GELU, no row statistics) may be cut into column ranges along blockIdx.y: one trip per thread
    const int c_begin = blockIdx.y * nthr, c_end = gridDim.y > 1 ? min(n4, c_begin + nthr) : n4;
    for (int c = c_begin + tid; c < c_end; c += nthr) {
        const int n = c * 4;
        float4 x;
        if (p.mode != 2) {
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* src = p.part + (size_t)m * p.ldp + n;
            int s = 0;
            if (p.eager && p.ksplit >= 8) {           // 8 slabs in flight (one round trip), summed in the order of the loop below
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *(const float4*)(src + (size_t)u * sstride);
#pragma unroll
                for (int u = 0; u < 8; u += 4) {
                    y.x += (t[u].x + t[u + 1].x) + (t[u + 2].x + t[u + 3].x); y.y += (t[u].y + t[u + 1].y) + (t[u + 2].y + t[u + 3].y);
                    y.z += (t[u].z + t[u + 1].z) + (t[u + 2].z + t[u + 3].z); y.w += (t[u].w + t[u + 1].w) + (t[u + 2].w + t[u + 3].w);
                }
                s = 8;
            }
            for (; s + 4 <= p.ksplit; s += 4) {       // 4 slabs in flight: the loop is L2-latency bound
                const float4 t0 = *(const float4*)(src + (size_t)s * sstride);
                const float4 t1 = *(const float4*)(src + (size_t)(s + 1) * sstride);
                const float4 t2 = *(const float4*)(src + (size_t)(s + 2) * sstride);
                const float4 t3 = *(const float4*)(src + (size_t)(s + 3) * sstride);
                y.x += (t0.x + t1.x) + (t2.x + t3.x); y.y += (t0.y + t1.y) + (t2.y + t3.y);
                y.z += (t0.z + t1.z) + (t2.z + t3.z); y.w += (t0.w + t1.w) + (t2.w + t3.w);
            }
            for (; s < p.ksplit; ++s) {
                const float4 t = *(const float4*)(src + (size_t)s * sstride);
                y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
            }
            if (p.bias) { const float4 bb = ld_h4(p.bias + n); y.x += bb.x; y.y += bb.y; y.z += bb.z; y.w += bb.w; }
            y.x = r16(y.x); y.y = r16(y.y); y.z = r16(y.z); y.w = r16(y.w);        // the Linear's fp16 output
            if (p.mode == 1) {
                if (p.gelu_kind == 2) x = make_float4(gelu_tanh(y.x), gelu_tanh(y.y), gelu_tanh(y.z), gelu_tanh(y.w));
                else x = make_float4(gelu_erf(y.x), gelu_erf(y.y), gelu_erf(y.z), gelu_erf(y.w));
                st_h4(p.out + (size_t)m * p.ldo + n, x);
                continue;
            }
            const float4 xr = ld_h4(p.x + (size_t)m * p.ldx + n);
            x = make_float4(r16(xr.x + y.x), r16(xr.y + y.y), r16(xr.z + y.z), r16(xr.w + y.w));
            st_h4(p.x + (size_t)m * p.ldx + n, x);                                  // residual stream, in place
        } else {
            x = ld_h4(p.x + (size_t)m * p.ldx + n);
        }
        *(float4*)&row[n] = x;
        sum += (x.x + x.y) + (x.z + x.w);
    }
    if (p.mode == 1 || p.mode == 3) return;
    // ---- LayerNorm: two-pass statistics in fp32 (torch_model.py:25-27) --------------------------
    const float mean = block_sum(sum, red) / (float)p.N;
    float q = 0.f;
    for (int c = tid; c < n4; c += nthr) {                  // each thread re-reads what it wrote
        const float4 x = *(const float4*)&row[c * 4];
        const float a = x.x - mean, b = x.y - mean, cc = x.z - mean, d = x.w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
    }
    const float rstd = rsqrtf(block_sum(q, red) / (float)p.N + 1e-5f);
    for (int c = tid; c < n4; c += nthr) {
        const int n = c * 4;
        const float4 x = *(const float4*)&row[n];
        const float4 g = ld_h4(p.ln_g + n), bt = ld_h4(p.ln_b + n);
        st_h4(p.out + (size_t)m * p.ldo + n,
              make_float4((x.x - mean) * rstd * g.x + bt.x, (x.y - mean) * rstd * g.y + bt.y,
                          (x.z - mean) * rstd * g.z + bt.z, (x.w - mean) * rstd * g.w + bt.w));
    }
}

int launch_row_finish(const RowFinishParams& p, hipStream_t stream) {
    WM_REQUIRE(p.N <= ROW_MAX_N && p.N % 4 == 0, "row_finish: N=%d must be a multiple of 4, <= %d", p.N, ROW_MAX_N);
    WM_REQUIRE(p.M > 0, "row_finish: empty M");
    WM_REQUIRE(p.ldx % 4 == 0 && p.ldo % 4 == 0 && p.ldp % 4 == 0, "row_finish: leading dimensions must be multiples of 4");
    // Two ways to run the same arithmetic (results are identical bit for bit, so the choice may depend on the row count):
    // few rows (small batches, latency-bound): all 8 slabs in flight and the GELU mode -- which has no row statistics --
    // cut into column ranges, one trip per thread (B = 8: 2.80 -> 2.64 ms per decode step);
    // many rows (the big-batch decode loop runs these kernels next to a K/V stream that owns the HBM, and the step is
    // bound by THAT stream): the gentle form, which costs the stream less (B = 384: 18.05 vs 18.4 ms per step).
    RowFinishParams q = p;
    q.eager = p.M < 64;
    const int n4 = p.N / 4;
    const int threads = ROW_THREADS;
    dim3 grid(p.M);
    if (q.eager && p.mode == 1 && n4 > threads) grid.y = (n4 + threads - 1) / threads;
    hipLaunchKernelGGL(row_finish_kernel, grid, dim3(threads), 0, stream, q);
    WM_LAUNCH_CHECK(stream, "row_finish");
    return 0;
}

// ---- standalone LayerNorm, streaming form: one WAVE per row, the row in registers ----------------------------------
// The encoder runs 65 LayerNorms over 1500 x batch rows: with one 256-thread workgroup per row (row_finish_kernel, the
// row staged in LDS, two block-wide reductions) a launch over 864 000 rows took 2.4 ms against 0.8 ms of HBM time.  Here a
// wave holds its rows in registers and streams: 8-byte pieces, two rows in flight per wave, no LDS, no barrier.
// The arithmetic is row_finish_kernel's bit for bit, so the two are interchangeable: lane l plays that kernel's threads
// l, l + 64, l + 128, l + 192 (thread t owns the 4-element groups t, t + 256, ...), the four per-wave sums are reduced
// separately and added in wave order, exactly as block_sum does.
constexpr int LNS_G = 6;                       // groups per lane: N <= 1536
constexpr int LNS_ROWS = 2;                    // rows in flight per wave

__global__ __launch_bounds__(256) void layernorm_stream_kernel(const h16* x, int ldx, int M, int N, const h16* g, const h16* b,
                                                               h16* out, int ldo) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n4 = N >> 2;
    const int row0 = (blockIdx.x * 4 + wid) * LNS_ROWS;
    if (row0 >= M) return;                     // wave-uniform
    // group j of this lane: column group c = lane + 64 * (j & 3) + 256 * (j >> 2)  (thread t = lane + 64 * (j & 3), trip j >> 2)
    half4v xv[LNS_ROWS][LNS_G], gv[LNS_G], bv[LNS_G];
#pragma unroll
    for (int r = 0; r < LNS_ROWS; ++r) {
        const h16* row = x + (size_t)min(row0 + r, M - 1) * ldx;
#pragma unroll
        for (int j = 0; j < LNS_G; ++j) {
            const int c = min(lane + 64 * (j & 3) + 256 * (j >> 2), n4 - 1);
            xv[r][j] = *(const half4v*)(row + c * 4);
        }
    }
#pragma unroll
    for (int j = 0; j < LNS_G; ++j) {
        const int c = min(lane + 64 * (j & 3) + 256 * (j >> 2), n4 - 1);
        gv[j] = *(const half4v*)(g + c * 4);
        bv[j] = *(const half4v*)(b + c * 4);
    }
#pragma unroll
    for (int r = 0; r < LNS_ROWS; ++r) {
        if (row0 + r >= M) break;              // wave-uniform, after every load was issued
        float xf[LNS_G][4];
#pragma unroll
        for (int j = 0; j < LNS_G; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) xf[j][e] = (float)xv[r][j][e];
        // pass 1: thread t's partial sum over its trips, per "wave" w = j & 3; then the four wave sums in order
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < (LNS_G + 3) / 4; ++k) {
                const int j = w + 4 * k;
                if (j < LNS_G && lane + 64 * w + 256 * k < n4) s += (xf[j][0] + xf[j][1]) + (xf[j][2] + xf[j][3]);
            }
            tot += wave_sum_nomfma(s);
        }
        const float mean = tot / (float)N;
        float qt = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < (LNS_G + 3) / 4; ++k) {
                const int j = w + 4 * k;
                if (j < LNS_G && lane + 64 * w + 256 * k < n4) {
                    const float a = xf[j][0] - mean, bb = xf[j][1] - mean, cc = xf[j][2] - mean, d = xf[j][3] - mean;
                    q += (a * a + bb * bb) + (cc * cc + d * d);
                }
            }
            qt += wave_sum_nomfma(q);
        }
        const float rstd = rsqrtf(qt / (float)N + 1e-5f);
        h16* orow = out + (size_t)(row0 + r) * ldo;
#pragma unroll
        for (int j = 0; j < LNS_G; ++j) {
            const int c = lane + 64 * (j & 3) + 256 * (j >> 2);
            if (c < n4)
                st_h4(orow + c * 4, make_float4((xf[j][0] - mean) * rstd * (float)gv[j][0] + (float)bv[j][0],
                                                (xf[j][1] - mean) * rstd * (float)gv[j][1] + (float)bv[j][1],
                                                (xf[j][2] - mean) * rstd * (float)gv[j][2] + (float)bv[j][2],
                                                (xf[j][3] - mean) * rstd * (float)gv[j][3] + (float)bv[j][3]));
        }
    }
}

int launch_layernorm(const h16* x, int ldx, int M, int N, const h16* g, const h16* b, h16* out, int ldo,
                     hipStream_t stream) {
    const char* form = lab_env_str("WM_LN_FORM");  // "workgroup": row_finish_kernel's form for every width (WM_LAB=1: A/B runs, the bit-identity test)
    if (N % 4 == 0 && N <= LNS_G * 256 && ldx % 4 == 0 && ldo % 4 == 0 && M > 0 && !(form && form[0] == 'w')) {      // by row width only: a row's result never depends on M
        const int rows_per_wg = 4 * LNS_ROWS;
        hipLaunchKernelGGL(layernorm_stream_kernel, dim3((M + rows_per_wg - 1) / rows_per_wg), dim3(256), 0, stream, x, ldx, M, N, g, b, out, ldo);
        WM_LAUNCH_CHECK(stream, "layernorm");
        return 0;
    }
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
    const int T = p.t_dev ? *p.t_dev : 0;            // graph replay: column / position offset read on the device
    // the step's generation number for the one-row chain's granule epochs (gemv_chain.hip): this kernel opens every decoder call,
    // every chain of the call starts behind it
    if (p.generation && m == 0 && threadIdx.x == 0) *p.generation += 1u;
    int tok = p.tokens[(size_t)(m / p.L) * p.tokens_ld + m % p.L + T];
    if (tok < 0) tok = 0;
    if (tok >= p.n_vocab) tok = p.n_vocab - 1;
    const int kt_total = p.C / 32;
    const unsigned char* base = (const unsigned char*)p.emb_tiles + (size_t)(tok >> 4) * kt_total * 1024 + (tok & 15) * 16;
    const h16* pos = p.pos + (size_t)(m % p.L + T) * p.C;
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
