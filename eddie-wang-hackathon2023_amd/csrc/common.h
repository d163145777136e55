// Shared device/host helpers for the gfx950 Whisper engine.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stddef.h>

namespace wm {

typedef _Float16 h16;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef short short4r __attribute__((__vector_size__(4 * sizeof(short))));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// ---- status / last error (thread local) ------------------------------------------------
void set_error(const char* fmt, ...);
#define WM_CHECK_HIP(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            wm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                          __LINE__);                                                        \
            return 2;                                                                       \
        }                                                                                   \
    } while (0)
#define WM_REQUIRE(cond, ...)                                                               \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            wm::set_error(__VA_ARGS__);                                                     \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)
// debug: WM_SYNC_CHECK=1 synchronises and checks after every launch (the reference's
// sync_check_cuda_error, cudaUtils.h:103-134)
int post_launch_check(hipStream_t s, const char* what);
#define WM_LAUNCH_CHECK(stream, what)                                                       \
    do {                                                                                    \
        int _r = wm::post_launch_check(stream, what);                                       \
        if (_r) return _r;                                                                  \
    } while (0)

// ---- lab knobs ---------------------------------------------------------------------------------
// Environment variables that change a schedule or the order of fp32 sums exist for A/B runs only (scripts/).  The
// library honours them ONLY when WM_LAB=1 is in the environment too, says so once per knob on stderr and lists them through
// wm_lab_knobs() (bench.py echoes that list in its line); without WM_LAB=1 a knob that is set is ignored, with one warning.
// Call sites cache the value in a function-local static: a knob is read once per process.  (WM_SYNC_CHECK, a debugging aid that
// changes no result, does not need WM_LAB.)
int lab_env_int(const char* name, int dflt);
const char* lab_env_str(const char* name);          // nullptr unless set and WM_LAB=1

// ---- device helpers ------------------------------------------------------------------
// Wave-wide reductions as an xor butterfly from 32 down to 1, entirely inside the ALU: v_permlane32_swap / v_permlane16_swap
// (gfx950) for the two cross-row steps, DPP row rotations and quad permutations for the rest.  __shfl_xor compiles to
// ds_bpermute_b32 plus a wait on the LDS counter per step -- six dependent LDS round trips per reduction, most of the time of
// the small kernels' LayerNorm / softmax tails.  The values added are the ones the xor butterfly adds, in its order (after
// the steps 32, 16, 8 a lane's value depends on lane mod 8 only, so the lane a rotation by 4 reaches holds exactly what lane ^ 4
// holds; likewise for 2 and 1 with the quad permutations): results are bit-identical to the __shfl_xor form.
// a * b rounded to fp32 ON ITS OWN: the statement keeps the compiler from fusing the product into a following addition (hipcc
// contracts by default and decides per context -- two copies of one expression can round differently; the merges of the
// cross-attention's key-range pieces exist in three places that must agree bit for bit)
__device__ __forceinline__ float mul_rn(float a, float b) {
    float r = a * b;
    asm volatile("" : "+v"(r));
    return r;
}
// An fp32 value as it stands, ahead of a conversion to fp16: without the statement the compiler may fold the conversion into the
// operation that produced the value (v_fma_mixlo_f16 rounds a * b + c ONCE, straight to fp16, instead of to fp32 and then to fp16)
// -- in one copy of a loop and not in another, depending on what it knows about the trip count.
__device__ __forceinline__ float f32_as_is(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
template <int CTRL>
__device__ __forceinline__ float wave_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// The two swap steps are written in assembly: the instruction exchanges halves BETWEEN two registers and has two results, and
// the compiler's handling of the second result of its intrinsic was wrong in context (v = r0 + r0).  x keeps its lower rows /
// lower half twice, y ends up with the upper ones twice: x + y is own + partner on every lane, in either order.
// The leading s_nops: the hazard recognizer does not look inside inline assembly, and the scratch register the move writes may
// still be the destination or the C operand of a matrix instruction in flight (found in the encoder attention kernel: wrong
// sums with fewer than five wait states there).  LLVM's gfx940/950 tables ask for up to 18 wait states between a 16-pass XDL
// write / SrcC read and a VALU write of the same register, so the helpers wait 19 (s_nop 7 + 7 + 2): safe wherever they are
// inlined, MFMA kernels included -- 11 more idle cycles per reduction, nothing next to the memory round trips these kernels
// are made of.  No kernel needs the guarded forms today: gemv_small / gemm_rows reduce in their LayerNorm prologues, before the
// wave's first MFMA (wave_sum_pre_mfma), rowops, greedy and attn_decode have no MFMA at all (wave_sum_nomfma / wave_max_nomfma:
// the same butterfly without the wait states); attn_encoder keeps __shfl_xor.  The guarded forms stay for any reduction that
// does follow an MFMA.
#define WM_SWAP_GUARD "s_nop 7\n\ts_nop 7\n\ts_nop 2\n\t"
__device__ __forceinline__ void wave_swap32(int& x, int& y) {
    asm volatile(WM_SWAP_GUARD "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "=&v"(y));
}
__device__ __forceinline__ void wave_swap16(int& x, int& y) {
    asm volatile(WM_SWAP_GUARD "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "=&v"(y));
}
// single butterfly steps across rows (lane ^ 16, lane ^ 32): own + partner / max(own, partner) on every lane
__device__ __forceinline__ float wave_add_xor16(float v) { int x = __builtin_bit_cast(int, v), y; wave_swap16(x, y); return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y); }
__device__ __forceinline__ float wave_add_xor32(float v) { int x = __builtin_bit_cast(int, v), y; wave_swap32(x, y); return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y); }
__device__ __forceinline__ float wave_max_xor16(float v) { int x = __builtin_bit_cast(int, v), y; wave_swap16(x, y); return fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y)); }
__device__ __forceinline__ float wave_max_xor32(float v) { int x = __builtin_bit_cast(int, v), y; wave_swap32(x, y); return fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y)); }
__device__ __forceinline__ float wave_sum(float v) {
    int x = __builtin_bit_cast(int, v), y;
    wave_swap32(x, y);
    v = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
    x = __builtin_bit_cast(int, v);
    wave_swap16(x, y);
    v = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
    v += wave_dpp<0x128>(v);            // row_ror:8   = lane ^ 8
    v += wave_dpp<0x124>(v);            // row_ror:4   ~ lane ^ 4
    v += wave_dpp<0x4E>(v);             // quad_perm [2,3,0,1] = lane ^ 2
    v += wave_dpp<0xB1>(v);             // quad_perm [1,0,3,2] = lane ^ 1
    return v;
}
// The same butterfly without the leading wait states, for kernels in which NO matrix instruction of the wave can be in flight at
// the call: kernels without MFMAs, or reductions that all precede the wave's first MFMA (gemm_rows.hip's LayerNorm prologue:
// two reductions per row and 8 rows per wave -- the 38 idle cycles per reduction were a sixth of that prologue).  Same values,
// same order: bit-identical to wave_sum.
__device__ __forceinline__ float wave_sum_pre_mfma(float v) {
    int x = __builtin_bit_cast(int, v), y;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "=&v"(y));
    v = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
    x = __builtin_bit_cast(int, v);
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "=&v"(y));
    v = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
    v += wave_dpp<0x128>(v);
    v += wave_dpp<0x124>(v);
    v += wave_dpp<0x4E>(v);
    v += wave_dpp<0xB1>(v);
    return v;
}
__device__ __forceinline__ float wave_sum_nomfma(float v) { return wave_sum_pre_mfma(v); }     // kernels without matrix instructions
// EIGHT values at a time through one cross-row step (own + partner of lane ^ 16 | lane ^ 32), for waves with no matrix instruction
// in flight: the eight moves, ONE pair of wait states, the eight swaps, one more pair -- instead of eight guarded single steps
// (wave_add_xor16: 19 + 4 wait states and a dependent move / swap / add each; volatile assembly statements keep their order, so
// eight of them never interleave).  Per value the same two operands are added in the same order: bit-identical.  The per-item tail
// of the decode cross-attention -- 8 or 16 partial sums per lane through two such steps -- took 0.78 us of the one-launch step's
// cross-attention stage with the single steps (scripts/lab/chain_stamps.py, profiles/r5q_*).
#define WM_X8_STEP(SWAP)                                                                                                         \
    int x0 = __builtin_bit_cast(int, v[0]), x1 = __builtin_bit_cast(int, v[1]), x2 = __builtin_bit_cast(int, v[2]),              \
        x3 = __builtin_bit_cast(int, v[3]), x4 = __builtin_bit_cast(int, v[4]), x5 = __builtin_bit_cast(int, v[5]),              \
        x6 = __builtin_bit_cast(int, v[6]), x7 = __builtin_bit_cast(int, v[7]), y0, y1, y2, y3, y4, y5, y6, y7;                  \
    asm volatile("v_mov_b32 %8, %0\n\tv_mov_b32 %9, %1\n\tv_mov_b32 %10, %2\n\tv_mov_b32 %11, %3\n\t"                            \
                 "v_mov_b32 %12, %4\n\tv_mov_b32 %13, %5\n\tv_mov_b32 %14, %6\n\tv_mov_b32 %15, %7\n\ts_nop 1\n\t"                   \
                 SWAP " %0, %8\n\t" SWAP " %1, %9\n\t" SWAP " %2, %10\n\t" SWAP " %3, %11\n\t"                                    \
                 SWAP " %4, %12\n\t" SWAP " %5, %13\n\t" SWAP " %6, %14\n\t" SWAP " %7, %15\n\ts_nop 1"                            \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),                               \
                   "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(y4), "=&v"(y5), "=&v"(y6), "=&v"(y7));                      \
    v[0] = __builtin_bit_cast(float, x0) + __builtin_bit_cast(float, y0); v[1] = __builtin_bit_cast(float, x1) + __builtin_bit_cast(float, y1);   \
    v[2] = __builtin_bit_cast(float, x2) + __builtin_bit_cast(float, y2); v[3] = __builtin_bit_cast(float, x3) + __builtin_bit_cast(float, y3);   \
    v[4] = __builtin_bit_cast(float, x4) + __builtin_bit_cast(float, y4); v[5] = __builtin_bit_cast(float, x5) + __builtin_bit_cast(float, y5);   \
    v[6] = __builtin_bit_cast(float, x6) + __builtin_bit_cast(float, y6); v[7] = __builtin_bit_cast(float, x7) + __builtin_bit_cast(float, y7);
__device__ __forceinline__ void wave_add_xor16_x8_nomfma(float* v) { WM_X8_STEP("v_permlane16_swap_b32") }
__device__ __forceinline__ void wave_add_xor32_x8_nomfma(float* v) { WM_X8_STEP("v_permlane32_swap_b32") }
#undef WM_X8_STEP
__device__ __forceinline__ float wave_max_nomfma(float v) {
    int x = __builtin_bit_cast(int, v), y;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "=&v"(y));
    v = fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
    x = __builtin_bit_cast(int, v);
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "=&v"(y));
    v = fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
    v = fmaxf(v, wave_dpp<0x128>(v));
    v = fmaxf(v, wave_dpp<0x124>(v));
    v = fmaxf(v, wave_dpp<0x4E>(v));
    v = fmaxf(v, wave_dpp<0xB1>(v));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    int x = __builtin_bit_cast(int, v), y;
    wave_swap32(x, y);
    v = fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
    x = __builtin_bit_cast(int, v);
    wave_swap16(x, y);
    v = fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y));
    v = fmaxf(v, wave_dpp<0x128>(v));
    v = fmaxf(v, wave_dpp<0x124>(v));
    v = fmaxf(v, wave_dpp<0x4E>(v));
    v = fmaxf(v, wave_dpp<0xB1>(v));
    return v;
}
// v_permlane16_swap: the odd 16-lane rows of a and the even rows of b change places (lane l of row 2k+1 <-> lane l of row 2k).
// For data that no matrix instruction of the wave has in flight (freshly converted values in an epilogue).
__device__ __forceinline__ void lane_rows_swap16(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float r16(float x) { return (float)(h16)x; }   // round through fp16

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. far below an fp16 ulp of any GELU output that matters) on the
// hardware reciprocal and exp2 (v_rcp_f32 / v_exp_f32, 1 ulp each -- inside the formula's own error): 14 instructions + 2
// transcendentals per value.  (The first version called __frcp_rn and __expf: a correctly rounded reciprocal is a full IEEE
// division sequence and __expf its own range reduction -- 35 instructions per value, found in the ISA of the mlp1 GEMM, whose
// whole deficit against the plain GEMMs was this epilogue.)
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = __builtin_amdgcn_exp2f(ax * (ax * -1.4426950408889634f));        // exp(-ax^2)
    return copysignf(1.0f - poly * e, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
// The same arithmetic on two values at a time: the multiplies and fused multiply-adds become packed fp32 instructions
// (v_pk_mul_f32 / v_pk_fma_f32: two values per issue), rcp and exp2 stay one per value.  Every operation is the scalar
// function's, in its order, so the results are bit-identical (scripts/lab/gemm_lab3.hip compares a GEMM that uses this
// form with one that uses the scalar form, element by element).  Round 4 re-measured the scalar form in the persistent GEMM's
// GELU epilogue (no SLP re-packing: 1 793 plain + 256 transcendental instructions per wave and tile against 896 packed + 256,
// and 74 instead of 335 hazard s_nops): 892-893 against 895-900 TFLOP/s on mlp1 -- the packed form stays
// (profiles/r4a_gemm_gelu_scalar_ab.log).
__device__ __forceinline__ float2v gelu_erf2(float2v x) {
    const float2v xs = x * 0.70710678118654752f;
    const float2v ax = {fabsf(xs[0]), fabsf(xs[1])};
    const float2v d = 1.0f + 0.3275911f * ax;
    const float2v t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const float2v poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float2v a2 = ax * (ax * -1.4426950408889634f);
    const float2v e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
    const float2v m = 1.0f - poly * e;
    const float2v erf = {copysignf(m[0], xs[0]), copysignf(m[1], xs[1])};
    return 0.5f * x * (1.0f + erf);
}
__device__ __forceinline__ float gelu_tanh(float x) {
    return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

// 4 signed int8 packed in a dword -> 4 fp16 (exact).  s8 ^ 0x80 = u8 in [0,255];
// 0x6400 | u8 is the fp16 number 1024 + u8; subtracting 1152 gives the signed value.
__device__ __forceinline__ void cvt_s8x4_f16x4(uint32_t w, half2v& lo, half2v& hi) {
    uint32_t u = w ^ 0x80808080u;
    uint32_t a = __builtin_amdgcn_perm(0x64646464u, u, 0x07010700u);   // bytes: [64,u1,64,u0]
    uint32_t b = __builtin_amdgcn_perm(0x64646464u, u, 0x07030702u);   // bytes: [64,u3,64,u2]
    const half2v k = {(h16)1152.0f, (h16)1152.0f};
    lo = __builtin_bit_cast(half2v, a) - k;
    hi = __builtin_bit_cast(half2v, b) - k;
}

}  // namespace wm
