// The fused Linear epilogue of the small-batch decode path (gemv_small.hip): what the big-batch path's row kernel does
// after the K slices are added, with the same rounding points.
#pragma once
#include "common.h"

namespace wm {

struct FusedEpilogue {
    int mode;                    // 0: fp32 sums -> out32   1: fp16(gelu(fp16(y + bias))) -> out16
                                 // 2: x = fp16(x + fp16(y + bias)) in place   3: fp16(y) -> out16, col < n_valid
    const h16* bias; int gelu_kind;
    float* out32; int ld32;
    h16* out16; int ld16; int n_valid;
    h16* x; int ldx;
};

// What the epilogue reads from memory besides the sums -- the channel's bias and, in mode 2, the tile's values of the residual
// stream -- requested by the caller at the START of the kernel (fused_epilogue_prefetch) instead of behind the barrier that joins
// the K slices: there they were a memory round trip of their own at the end of every launch.
// The loads are UNCONDITIONAL (a mode that has no bias / no residual reads a few bytes of `any`, a valid address, and drops them):
// behind a run-time test hipcc branches around a load and waits for it inside the branch.
struct FusedEpiloguePre { h16 bias_raw; bool has_bias; h16 x[4]; __device__ float bias() const { return has_bias ? (float)bias_raw : 0.f; } };
__device__ __forceinline__ FusedEpiloguePre fused_epilogue_prefetch(const FusedEpilogue& e, int M, int nb, int mt, int lane, const void* any) {
    const int g = lane >> 4, col = nb * 16 + (lane & 15);
    FusedEpiloguePre pre;
    pre.has_bias = (e.mode == 1 || e.mode == 2) && e.bias != nullptr;
    pre.bias_raw = *((pre.has_bias ? e.bias : (const h16*)any) + (pre.has_bias ? col : 0));
    const bool add_x = e.mode == 2;
    const h16* xs = add_x ? e.x + col : (const h16*)any;
    const size_t xld = add_x ? (size_t)e.ldx : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = min(mt * 16 + g * 4 + r, M - 1);                 // (rows past the end re-read the last one; never used)
        pre.x[r] = xs[(size_t)row * xld];
    }
    return pre;
}

// One 16 x 16 accumulator tile (MFMA C/D layout: lane -> channel nb * 16 + (lane & 15), rows mt * 16 + 4 (lane >> 4) + r)
// whose values y[r] are the Linear's fp32 sums (scaled, K slices already combined).  pre: the tile's prefetched bias / residual
// values, or nullptr (they are read here then).
__device__ __forceinline__ void fused_epilogue_tile(const FusedEpilogue& e, int M, int nb, int mt, int lane, const float (&y)[4],
                                                    const FusedEpiloguePre* pre = nullptr) {
    const int g = lane >> 4, col = nb * 16 + (lane & 15);
    const float bias = pre ? pre->bias() : ((e.mode == 1 || e.mode == 2) && e.bias ? (float)e.bias[col] : 0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + g * 4 + r;
        if (row >= M) continue;
        if (e.mode == 0) {                             // raw sums for the attention kernels (they add the bias and round)
            e.out32[(size_t)row * e.ld32 + col] = y[r];
        } else if (e.mode == 3) {                      // logits: fp16(x . E^T), whisper/model.py:288-290
            if (col < e.n_valid) e.out16[(size_t)row * e.ld16 + col] = (h16)y[r];
        } else {
            const float y16 = r16(y[r] + bias);                                       // the Linear's fp16 output
            if (e.mode == 1) {
                e.out16[(size_t)row * e.ld16 + col] = (h16)(e.gelu_kind == 2 ? gelu_tanh(y16) : gelu_erf(y16));
            } else {                                   // mode 2: residual stream, in place
                h16* xp = e.x + (size_t)row * e.ldx + col;
                const h16 x_old = pre ? pre->x[r] : *xp;
                *xp = (h16)r16((float)x_old + y16);
            }
        }
    }
}

}  // namespace wm
