// Device-side greedy step: Whisper's logit filters + argmax + log-prob bookkeeping + token append
// in ONE kernel, so that a decode step never returns to the host.
//
// Restates for the device what the reference does in Python on every token (host loops,
// .tolist() and a stream sync per step -- W/decoding.py:785-821):
//   SuppressBlank (W/decoding.py:202-209), SuppressTokens (:212-217, list from :394-421),
//   ApplyTimestampRules (:134-199), GreedyDecoder.update (:274-300).
// One workgroup per utterance; two passes over the vocabulary row (51 865 fp16 logits = 104 KB,
// L2-resident right after the logits GEMM):
//   pass 1: masked running (max, sum-exp) separately for text tokens (< timestamp_begin) and
//           timestamp tokens, plus the arg-max of each class;
//   decide: timestamps win if logsumexp(timestamps) > max(text)  (:191-199; the common log Z
//           cancels), then the next token, its log-probability under the final mask, EOT
//           stickiness and the append.
// temperature > 0 (round 4): the reference draws next ~ Categorical(logits / T) on the host (GreedyDecoder.update, :282-285) and
// books log_softmax(logits)[next] at T = 1.  Here the draw is a Gumbel-max: next = argmax_n (x_n / T + g_n), g_n = -log(-log u_n),
// which has exactly that distribution; u_n comes from a counter-based generator keyed on (seed, global row, position, token), so
// a draw depends on nothing but its own coordinates (not on the batch, the utterance groups or the launch shape), the same
// kernel serves every step of a replayed graph (the seed sits in device memory) and the host can recompute any draw
// (tests/test_gpu_round4.py).  The draws cannot equal torch's generator's: the distribution, the masks and the log-probability
// bookkeeping are what the tests hold to the reference.
#include "common.h"
#include "kernels.h"

namespace wm {

struct MS { float m, s; };                       // running max and sum of exp(x - m)
__device__ __forceinline__ MS ms_add(MS a, float x) {
    if (x == -INFINITY) return a;
    if (x > a.m) { a.s = a.s * __expf(a.m - x) + 1.f; a.m = x; }
    else a.s += __expf(x - a.m);
    return a;
}
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
    if (b.m == -INFINITY) return a;
    if (a.m == -INFINITY) return b;
    const float m = fmaxf(a.m, b.m);
    return MS{m, a.s * __expf(a.m - m) + b.s * __expf(b.m - m)};
}
struct AM { float v; int i; };                   // arg-max with first-index tie break
__device__ __forceinline__ AM am_merge(AM a, AM b) {
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

template <typename T, typename F>
__device__ __forceinline__ T wave_reduce(T v, F merge) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T other;
        // shuffle the struct field-wise (two 32-bit words)
        static_assert(sizeof(T) == 8, "pair types only");
        unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
        unsigned lo = __shfl_xor((unsigned)bits, o), hi = __shfl_xor((unsigned)(bits >> 32), o);
        other = __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
        v = merge(v, other);
    }
    return v;
}
template <typename T, typename F>
__device__ __forceinline__ T block_reduce(T v, F merge, T* scratch) {
    v = wave_reduce(v, merge);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T r = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = merge(r, scratch[w]);
    __syncthreads();
    return r;
}

constexpr int GREEDY_THREADS = 1024;

// uniform in (0, 1) from (seed, row, position, token): three rounds of the murmur3 finaliser over the mixed-in coordinates
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float gumbel_noise(uint32_t key, int n) {
    const uint32_t h = fmix32(fmix32(key ^ ((uint32_t)n * 0x27d4eb2fu)) + 0x9e3779b9u * (uint32_t)n);
    const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1), 24 bits
    return -__logf(-__logf(u));
}

__global__ __launch_bounds__(GREEDY_THREADS) void greedy_kernel(GreedyParams p) {
    __shared__ MS s_ms[2][GREEDY_THREADS / 64];
    __shared__ AM s_am[4][GREEDY_THREADS / 64];
    __shared__ int s_info[4], s_hist[GREEDY_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cur_len = p.t_dev ? *p.t_dev + 1 : p.cur_len;     // device step counter holds n_past = cur_len - 1
    h16* lg = p.logits + (size_t)b * p.ld_row;
    int32_t* toks = p.tokens + (size_t)b * p.ld_tok;
    const int tb = p.timestamp_begin;
    // apply_rules: 0 plain arg-max; 1 SuppressBlank + SuppressTokens + ApplyTimestampRules (the default decoding options);
    // 2 the two suppress filters WITHOUT the timestamp rules -- DecodingOptions.without_timestamps, where the reference builds no
    // ApplyTimestampRules filter (W/decoding.py:337-346) and samples from the whole (suppressed) vocabulary
    const bool lists = p.apply_rules != 0, ts_rules = p.apply_rules == 1;
    const bool first = lists && (cur_len == p.sample_begin);

    // ---- SuppressTokens (+ no_timestamps) and SuppressBlank: written into the logits row, exactly
    // like the reference's in-place filters (decoding.py:202-217); the scan below then sees -inf ----
    if (lists) {
        const h16 ninf = (h16)(-INFINITY);
        for (int i = tid; i < p.n_suppress; i += GREEDY_THREADS) lg[p.suppress[i]] = ninf;
        if (first) for (int i = tid; i < p.n_blank; i += GREEDY_THREADS) lg[p.blank[i]] = ninf;
    }
    // ---- token-history facts: last / penultimate sampled token, last timestamp.  Every thread looks at one sampled token
    // (a backwards scan by one thread was a chain of dependent loads: up to one L2 round trip per sampled token) ------------
    const int n_sampled = cur_len - p.sample_begin;
    int my_rel = -1, my_tok = -1;                  // this thread's latest timestamp among the sampled tokens it looked at
    if (ts_rules) {
        for (int j = tid; j < n_sampled; j += GREEDY_THREADS) {
            const int t = toks[p.sample_begin + j];
            if (t >= tb) { my_rel = j; my_tok = t; }
            if (j == n_sampled - 1) s_info[0] = t >= tb;
            if (j == n_sampled - 2) s_info[1] = t >= tb;
        }
    }
    {
        int r = my_rel;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r = max(r, __shfl_xor(r, o));
        if ((tid & 63) == 0) s_hist[tid >> 6] = r;
    }
    __syncthreads();                       // also orders the -inf stores above before the scan
    int rel_last = -1;
#pragma unroll
    for (int w = 0; w < GREEDY_THREADS / 64; ++w) rel_last = max(rel_last, s_hist[w]);
    if (rel_last >= 0 && my_rel == rel_last) s_info[2] = my_tok;          // exactly one thread holds that position
    __syncthreads();
    const bool last_ts = ts_rules && n_sampled >= 1 && s_info[0];
    const bool pen_ts = ts_rules && (n_sampled < 2 || s_info[1]);
    int ts_last = rel_last >= 0 ? s_info[2] : -1;
    if (ts_last >= 0 && !(last_ts && !pen_ts)) ts_last += 1;

    // allowed = [lo_txt, hi_txt) U [lo_ts, hi_ts): every rule of ApplyTimestampRules is a range
    int lo_txt = 0, hi_txt = ts_rules ? tb : p.V, lo_ts = ts_rules ? tb : p.V, hi_ts = p.V;
    if (ts_rules) {
        if (first) { hi_txt = 0; if (p.max_initial_ts >= 0) hi_ts = min(hi_ts, tb + p.max_initial_ts + 1); }
        if (last_ts) { if (pen_ts) lo_ts = p.V; else lo_txt = max(lo_txt, p.eot); }
        if (ts_last >= 0) lo_ts = max(lo_ts, ts_last);
    }

    MS txt{-INFINITY, 0.f}, tsm{-INFINITY, 0.f};
    AM atxt{-INFINITY, 0x7fffffff}, ats{-INFINITY, 0x7fffffff};
    // sampling: arg-max of the perturbed logits per class (wave-uniform switch; the greedy path is untouched)
    const bool sampling = p.temperature > 0.f;
    const float inv_temp = sampling ? 1.0f / p.temperature : 0.f;
    uint32_t rng_key = 0;
    if (sampling) {
        const uint32_t s_lo = p.seed_dev ? p.seed_dev[0] : p.seed_lo, s_hi = p.seed_dev ? p.seed_dev[1] : p.seed_hi;
        rng_key = fmix32(fmix32(s_lo ^ ((uint32_t)(p.row0 + b) * 0x9e3779b1u)) ^ s_hi ^ ((uint32_t)cur_len * 0x85ebca77u));
    }
    AM ptxt{-INFINITY, 0x7fffffff}, pts{-INFINITY, 0x7fffffff};
    auto perturb = [&](int n, float x) {                           // one token of an allowed class
        if (x == -INFINITY) return;
        const float y = x * inv_temp + gumbel_noise(rng_key, n);
        if (n < hi_txt) { if (y > ptxt.v) ptxt = AM{y, n}; }
        else if (y > pts.v) pts = AM{y, n};
    };
    auto visit = [&](int n, float x) {
        if (n < hi_txt) { if (n >= lo_txt) { txt = ms_add(txt, x); if (x > atxt.v) atxt = AM{x, n}; if (sampling) perturb(n, x); } }
        else if (n >= lo_ts && n < hi_ts) { tsm = ms_add(tsm, x); if (x > ats.v) ats = AM{x, n}; if (sampling) perturb(n, x); }
    };
    // eight logits that all belong to one class: one maximum, at most one rescale of the running sum, eight exponentials --
    // instead of eight dependent (compare, branch, exponential) updates
    auto visit8 = [&](MS& ms, AM& am, int n0, const float (&f)[8]) {
        const float m8 = fmaxf(fmaxf(fmaxf(f[0], f[1]), fmaxf(f[2], f[3])), fmaxf(fmaxf(f[4], f[5]), fmaxf(f[6], f[7])));
        if (m8 == -INFINITY) return;
        if (m8 > ms.m) { ms.s *= __expf(ms.m - m8); ms.m = m8; }         // (first finite value: 0 * exp(-inf) = 0)
        float e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = __expf(f[i] - ms.m);
        ms.s += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
        if (m8 > am.v) {
            int first_e = 7;
#pragma unroll
            for (int i = 6; i >= 0; --i) if (f[i] == m8) first_e = i;
            am = AM{m8, n0 + first_e};
        }
    };
    // rows are only 2-byte aligned (odd vocabulary): scalar head up to a 16-byte boundary, 8-wide body
    const int head = min(p.V, (int)(((16 - ((size_t)lg & 15)) & 15) >> 1));
    for (int n = tid; n < head; n += GREEDY_THREADS) visit(n, (float)lg[n]);
    const int nvec = (p.V - head) >> 3;
    for (int c = tid; c < nvec; c += GREEDY_THREADS) {
        const int n0 = head + c * 8;
        const half8v v = *(const half8v*)(lg + n0);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
        if (n0 >= lo_txt && n0 + 8 <= hi_txt) {
            visit8(txt, atxt, n0, f);
            if (sampling) {
#pragma unroll
                for (int e = 0; e < 8; ++e) perturb(n0 + e, f[e]);
            }
        } else if (n0 >= hi_txt && n0 >= lo_ts && n0 + 8 <= hi_ts) {
            visit8(tsm, ats, n0, f);
            if (sampling) {
#pragma unroll
                for (int e = 0; e < 8; ++e) perturb(n0 + e, f[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) visit(n0 + e, f[e]);
        }
    }
    for (int n = head + nvec * 8 + tid; n < p.V; n += GREEDY_THREADS) visit(n, (float)lg[n]);

    // the four results meet in one exchange: wave-level merges, one LDS round, wave 0 merges the sixteen partial results
    // (thread 0 alone uses them)
    txt = wave_reduce(txt, ms_merge); tsm = wave_reduce(tsm, ms_merge);
    atxt = wave_reduce(atxt, am_merge); ats = wave_reduce(ats, am_merge);
    if (sampling) { ptxt = wave_reduce(ptxt, am_merge); pts = wave_reduce(pts, am_merge); }
    if ((tid & 63) == 0) {
        s_ms[0][tid >> 6] = txt; s_ms[1][tid >> 6] = tsm; s_am[0][tid >> 6] = atxt; s_am[1][tid >> 6] = ats;
        if (sampling) { s_am[2][tid >> 6] = ptxt; s_am[3][tid >> 6] = pts; }
    }
    __syncthreads();
    if (tid >= 64) return;
    {
        constexpr int NW = GREEDY_THREADS / 64;
        txt = tid < NW ? s_ms[0][tid] : MS{-INFINITY, 0.f}; tsm = tid < NW ? s_ms[1][tid] : MS{-INFINITY, 0.f};
        atxt = tid < NW ? s_am[0][tid] : AM{-INFINITY, 0x7fffffff}; ats = tid < NW ? s_am[1][tid] : AM{-INFINITY, 0x7fffffff};
        txt = wave_reduce(txt, ms_merge); tsm = wave_reduce(tsm, ms_merge);
        atxt = wave_reduce(atxt, am_merge); ats = wave_reduce(ats, am_merge);
        if (sampling) {
            ptxt = tid < NW ? s_am[2][tid] : AM{-INFINITY, 0x7fffffff}; pts = tid < NW ? s_am[3][tid] : AM{-INFINITY, 0x7fffffff};
            ptxt = wave_reduce(ptxt, am_merge); pts = wave_reduce(pts, am_merge);
        }
    }

    if (tid == 0) {
        bool ts_only = false;
        if (ts_rules) {
            const float lse_ts = (tsm.m == -INFINITY) ? -INFINITY : tsm.m + __logf(tsm.s);
            ts_only = lse_ts > txt.m;      // logsumexp(ts logprobs) > max(text logprobs)
        }
        AM best = ts_only ? ats : am_merge(atxt, ats);
        if (sampling) {                 // the draw among the allowed tokens; its log-probability is the UNperturbed logit's (T = 1)
            best = ts_only ? pts : am_merge(ptxt, pts);
            if (best.i != 0x7fffffff) best.v = (float)lg[best.i];
        }
        MS z = ts_only ? tsm : ms_merge(txt, tsm);
        const float logz = z.m + logf(z.s);
        const float lp = best.v - logz;
        const int prev = toks[cur_len - 1];
        const bool alive = (prev != p.eot);
        // per-row sample_len: the row has sampled its quota -> it ends here, like the reference's loop at sample_len
        // (no log-probability for the closing EOT, W/decoding.py:794,817-821 + finalize :296-300)
        const bool capped = alive && p.row_limit && (cur_len - p.sample_begin >= p.row_limit[b]);
        if (alive && !capped) p.sum_logprobs[b] += lp;
        const int next = (alive && !capped) ? best.i : p.eot;
        toks[cur_len] = next;
        if (next == p.eot) {
            if (p.n_done) atomicAdd(p.n_done, 1);
            if (p.done) p.done[b] = 1;
        }
    }
}

int launch_greedy(const GreedyParams& p, hipStream_t stream) {
    WM_REQUIRE(p.t_dev || (p.cur_len >= 1 && p.cur_len < p.ld_tok), "greedy: cur_len=%d does not fit ld_tok=%d", p.cur_len, p.ld_tok);
    WM_REQUIRE(p.n_suppress == 0 || p.suppress != nullptr, "greedy: suppress list is null");
    hipLaunchKernelGGL(greedy_kernel, dim3(p.B), dim3(GREEDY_THREADS), 0, stream, p);
    WM_LAUNCH_CHECK(stream, "greedy");
    return 0;
}

// plain arg-max of fp16 rows, first index wins ties (torch.argmax's convention; the kernel-level entry wm_argmax)
__global__ __launch_bounds__(GREEDY_THREADS) void argmax_kernel(const h16* logits, long ld_row, int V, int32_t* ids) {
    __shared__ AM s_am[GREEDY_THREADS / 64];
    const h16* lg = logits + (size_t)blockIdx.x * ld_row;
    AM a{-INFINITY, 0x7fffffff};
    for (int n = threadIdx.x; n < V; n += GREEDY_THREADS) {
        const float x = (float)lg[n];
        if (x > a.v) a = AM{x, n};
    }
    a = block_reduce(a, am_merge, s_am);
    if (threadIdx.x == 0) ids[blockIdx.x] = a.i;
}

int launch_argmax(const h16* logits, long ld_row, int B, int V, int32_t* ids, hipStream_t stream) {
    hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(GREEDY_THREADS), 0, stream, logits, ld_row, V, ids);
    WM_LAUNCH_CHECK(stream, "argmax");
    return 0;
}

__global__ void step_advance_kernel(int32_t* counter) { if (threadIdx.x == 0 && blockIdx.x == 0) *counter += 1; }

// End of a group's decode step: advance the step counter and rebuild the list of rows still decoding (ascending order:
// ballot + prefix count per wave, wave offsets through LDS).  One workgroup, B <= 1024.
__global__ __launch_bounds__(1024) void step_finish_kernel(int32_t* counter, const int32_t* done, int B, int32_t* live) {
    __shared__ int s_cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0 && counter) *counter += 1;
    const bool alive = tid < B && done[tid] == 0;
    const unsigned long long mask = __ballot(alive);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[wid] = __popcll(mask);
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) { if (w < wid) base += s_cnt[w]; total += s_cnt[w]; }
    if (alive) live[1 + base + before] = tid;
    if (tid == 0) live[0] = total;
}

int launch_step_finish(int32_t* counter, const int32_t* done, int B, int32_t* live, hipStream_t stream) {
    WM_REQUIRE(done && live && B >= 1 && B <= 1024, "step_finish: null argument or batch %d outside [1, 1024]", B);
    hipLaunchKernelGGL(step_finish_kernel, dim3(1), dim3(1024), 0, stream, counter, done, B, live);
    WM_LAUNCH_CHECK(stream, "step_finish");
    return 0;
}

int launch_step_advance(int32_t* counter, hipStream_t stream) {
    WM_REQUIRE(counter, "step_advance: null counter");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, stream, counter);
    WM_LAUNCH_CHECK(stream, "step_advance");
    return 0;
}

}  // namespace wm
