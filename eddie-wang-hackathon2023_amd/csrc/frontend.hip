// Audio front end on the device: 16 kHz PCM -> Whisper log-mel spectrogram.
//
// Replaces log_mel_spectrogram (W/whisper_utils.py:99-146): torch.stft(n_fft 400, hop 160, periodic
// Hann window, centre + reflect padding) -> |.|^2 of the first n_frames frames -> mel projection ->
// log10(max(., 1e-10)) -> max(., clip_max - 8) -> (. + 4) / 4, one clip per batch row.
//
// n_fft = 400 is not a power of two and the whole transform is 0.97 GFLOP per 30 s clip (the encoder
// behind it is 2 272 GFLOP), so the DFT is evaluated directly in fp32 from an LDS twiddle table:
// one workgroup owns 32 frames (their windowed samples live in LDS and are read as wave-wide
// broadcasts), one thread owns one frequency bin for 8 frames at a time.  The power spectrum never
// leaves LDS; the mel projection, log10 and the per-clip maximum (order-independent, so an integer
// atomicMax on a monotone key is deterministic) are fused behind it.
#include "common.h"
#include "kernels.h"

namespace wm {

namespace {
constexpr int NFFT = 400, HOP = 160, NBIN = NFFT / 2 + 1, FT = 32, FB = 8;
constexpr int FW_LD = NFFT + 4;     // floats; rows stay 16-byte aligned for the float4 broadcasts
constexpr int P_LD = NBIN + 4;      // 205: odd stride -> the 32 frames of one bin hit 32 banks

__device__ __forceinline__ int float_key(float v) {       // monotone float -> int map
    const int b = __float_as_int(v);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }
}  // namespace

__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ audio, long audio_ld, int n_samples,
                                                       int n_frames, const float* __restrict__ filters, int n_mels,
                                                       float* __restrict__ logspec, int* __restrict__ gmax_key) {
    __shared__ __attribute__((aligned(16))) float fw[FT * FW_LD];
    __shared__ float2 tw[NFFT];
    __shared__ float win[NFFT];
    __shared__ float P[FT * P_LD];
    __shared__ float red[4];
    const int tid = threadIdx.x, b = blockIdx.y, f0 = blockIdx.x * FT;
    const float* clip = audio + (size_t)b * audio_ld;

    for (int j = tid; j < NFFT; j += 256) {
        const double a = 6.283185307179586476925 * (double)j / (double)NFFT;
        tw[j] = make_float2((float)cos(a), (float)sin(a));
        win[j] = (float)(0.5 - 0.5 * cos(a));                // torch.hann_window(400), periodic
    }
    __syncthreads();
    for (int i = tid; i < FT * NFFT; i += 256) {
        const int f = i / NFFT, n = i - f * NFFT;
        const int frame = f0 + f;
        float v = 0.f;
        if (frame < n_frames) {
            int s = frame * HOP + n - NFFT / 2;              // centre = True
            if (s < 0) s = -s;                               // pad_mode = 'reflect'
            if (s >= n_samples) s = 2 * (n_samples - 1) - s;
            v = clip[s] * win[n];
        }
        fw[f * FW_LD + n] = v;
    }
    __syncthreads();

    const int k = tid;
    if (k < NBIN) {
        for (int fb = 0; fb < FT; fb += FB) {
            float re[FB], im[FB];
#pragma unroll
            for (int j = 0; j < FB; ++j) re[j] = im[j] = 0.f;
            int idx = 0;                                     // (k * n) mod 400
            for (int n = 0; n < NFFT; n += 4) {
                float4 v[FB];
#pragma unroll
                for (int j = 0; j < FB; ++j) v[j] = *(const float4*)&fw[(fb + j) * FW_LD + n];
                float2 c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c[u] = tw[idx];
                    idx += k;
                    if (idx >= NFFT) idx -= NFFT;
                }
#pragma unroll
                for (int j = 0; j < FB; ++j) {
                    re[j] = fmaf(v[j].x, c[0].x, re[j]); im[j] = fmaf(v[j].x, c[0].y, im[j]);
                    re[j] = fmaf(v[j].y, c[1].x, re[j]); im[j] = fmaf(v[j].y, c[1].y, im[j]);
                    re[j] = fmaf(v[j].z, c[2].x, re[j]); im[j] = fmaf(v[j].z, c[2].y, im[j]);
                    re[j] = fmaf(v[j].w, c[3].x, re[j]); im[j] = fmaf(v[j].w, c[3].y, im[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < FB; ++j) P[(fb + j) * P_LD + k] = re[j] * re[j] + im[j] * im[j];
        }
    }
    __syncthreads();

    float lmax = -INFINITY;
    for (int o = tid; o < n_mels * FT; o += 256) {
        const int m = o / FT, f = o - m * FT;
        const float* frow = filters + (size_t)m * NBIN;
        const float* prow = P + f * P_LD;
        float acc = 0.f;
        for (int kk = 0; kk < NBIN; ++kk) acc = fmaf(frow[kk], prow[kk], acc);
        const int frame = f0 + f;
        if (frame < n_frames) {
            const float lg = log10f(fmaxf(acc, 1e-10f));
            logspec[((size_t)b * n_mels + m) * n_frames + frame] = lg;
            lmax = fmaxf(lmax, lg);
        }
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) {
        const float v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (v > -INFINITY) atomicMax(gmax_key + b, float_key(v));
    }
}

__global__ __launch_bounds__(256) void mel_finalize_kernel(const float* __restrict__ logspec, const int* __restrict__ gmax_key,
                                                           long per_clip, h16* __restrict__ out16, float* __restrict__ out32) {
    const int b = blockIdx.y;
    const float floor_v = key_float(gmax_key[b]) - 8.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_clip; i += (long)gridDim.x * 256) {
        const float v = (fmaxf(logspec[b * per_clip + i], floor_v) + 4.0f) / 4.0f;
        if (out32) out32[b * per_clip + i] = v;
        if (out16) out16[b * per_clip + i] = (h16)v;
    }
}

size_t log_mel_workspace_bytes(int batch, int n_samples, int n_mels) {
    const size_t n_frames = (size_t)n_samples / HOP;
    return (size_t)batch * n_mels * n_frames * sizeof(float) + (((size_t)batch * sizeof(int) + 255) & ~(size_t)255);
}

int launch_log_mel(const float* audio, int batch, int n_samples, long audio_ld, const float* filters, int n_mels,
                   h16* out16, float* out32, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    WM_REQUIRE(batch >= 1 && n_samples >= NFFT, "log_mel: batch=%d n_samples=%d", batch, n_samples);
    WM_REQUIRE(n_samples % HOP == 0, "log_mel: n_samples=%d must be a multiple of the hop (%d)", n_samples, HOP);
    WM_REQUIRE(n_mels >= 1 && filters != nullptr, "log_mel: mel filterbank missing");
    WM_REQUIRE(out16 != nullptr || out32 != nullptr, "log_mel: no output buffer");
    WM_REQUIRE(workspace_bytes >= log_mel_workspace_bytes(batch, n_samples, n_mels), "log_mel: workspace too small (%zu B)",
               workspace_bytes);
    const int n_frames = n_samples / HOP;
    float* logspec = (float*)workspace;
    int* gmax = (int*)((char*)workspace + (size_t)batch * n_mels * n_frames * sizeof(float));
    WM_CHECK_HIP(hipMemsetAsync(gmax, 0x80, (size_t)batch * sizeof(int), stream));    // key 0x80808080: below every float
    hipLaunchKernelGGL(stft_mel_kernel, dim3((n_frames + FT - 1) / FT, batch), dim3(256), 0, stream, audio, audio_ld,
                       n_samples, n_frames, filters, n_mels, logspec, gmax);
    WM_LAUNCH_CHECK(stream, "stft_mel");
    const long per_clip = (long)n_mels * n_frames;
    hipLaunchKernelGGL(mel_finalize_kernel, dim3((unsigned)min(256L, (per_clip + 255) / 256), batch), dim3(256), 0, stream,
                       logspec, gmax, per_clip, out16, out32);
    WM_LAUNCH_CHECK(stream, "mel_finalize");
    return 0;
}

}  // namespace wm
