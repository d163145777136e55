// Internal launch interfaces of the gfx950 kernels (C++ side of the C-ABI in include/whisper_mi355.h).
#pragma once
#include "common.h"

namespace wm {

// ---------------------------------------------------------------- gemm_f16.hip
struct GemmBigParams {
    const h16* A; int lda; int M; int K;
    const void* W; int N;                    // W [N][K] row-major fp16 (int8 matrices are expanded first: launch_dequant_w8)
    const h16* bias;                         // per output channel; may be null
    h16* C; int ldc;
    const h16* residual; int ldr; int res_mod;   // v += residual[row % res_mod (or row)][col]
    int act;                                 // 0 none, 1 exact-erf GELU, 2 tanh GELU
    int colscale_n; float colscale;          // columns < colscale_n are multiplied by colscale
    int out_mode;                            // 0 row-major, 1 head-split [B,2,H,T,64]
    int hs_T, hs_H, hs_kv;                   // hs_kv < 0: N = 2*H*64, kv = col / (H*64)
    float q8_inv_scale;                      // > 0 (head-split only): C is int8, code = sat_s8(rne(fp16 result * q8_inv_scale))
    // batched strided views (convolutions as GEMMs over a zero-padded token-major buffer):
    // row m lives at A + (m / a_rows) * a_bstride + (m % a_rows) * lda   (a_rows == 0: plain)
    int a_rows; long a_bstride;
    int c_rows; long c_bstride;              // same for C (out_mode 0 only)
    // > 0: at most this many (persistent, one per CU) workgroups -- the launch then leaves the other CUs to whatever runs
    // beside it (wm_encoder_forward_shared); the tiles and their arithmetic are the same, only who computes them changes
    int max_wgs;
    int nt_flags;                            // unused by the product kernels (the lab patch scripts/lab/gemm_f16p_stamps_whole_stage_nt.patch reads it: 1 = A panels by non-temporal DMA, 4 = non-temporal stores of C)
    int tile_rows;                           // persistent kernel: row panels per step of the tile order (0: chosen from K; 1: plain row-major)
};
int launch_gemm_f16(const GemmBigParams& p, hipStream_t stream);      // dispatch: gemm_f16p.hip when it supports the shape and the launch has enough tiles, else gemm_f16.hip's kernels
// launches with fewer 256 x 256 tiles than this run 128 x 128 tiles, two workgroups per CU (gemm_f16.hip; bit-identical results)
constexpr int GEMM_SMALL_TILES_DEFAULT = 150;
constexpr int GEMM_TINY_TILES_DEFAULT = 160;       // at most this many 128 x 128 tiles: 64 x 128 tiles instead (gemm_f16.hip)
void set_gemm_small_tiles(int tiles);     // 0: never the small form; < 0: the default
int get_gemm_small_tiles();
int launch_gemm_f16p(const GemmBigParams& p, hipStream_t stream);     // persistent 256x256 tiles, continuous LDS-DMA stream, alternating wave groups
bool gemm_f16p_supports(const GemmBigParams& p);
int launch_dequant_w8(const int8_t* q, const h16* scale, h16* out, int N, int K, hipStream_t stream);

// ---------------------------------------------------------------- gemm_skinny.hip
// Weight-streaming GEMM for M <= 64 rows (decode step M = B, prefill M = 3B):
//   out = A[M,K] x Wt     with Wt stored tile-linear (see weight layout in DESIGN.md)
struct GemmSkinnyParams {
    const h16* A; int lda; int M; int K;
    const void* Wt; int n_blocks; int w8;    // n_blocks = padded N / 16
    const h16* scale;                        // [n_blocks*16] (w8 only)
    int ksplit;                              // number of K slices
    float* part;                             // [ksplit][..][n_blocks*16] fp32 partial sums (scaled)
    long part_sstride;                       // elements between K-slice slabs (0: M * n_blocks * 16)
    h16* out; int ldc; int n_valid;          // direct fp16 output instead of partials (ksplit == 1)
};
constexpr int SKINNY_MAX_M = 256;   // rows per launch of the weight-streaming GEMM (16 MFMA row tiles)
int launch_gemm_skinny(const GemmSkinnyParams& p, hipStream_t stream);
int skinny_default_ksplit(int M, int K, int n_blocks, int w8);

// ---------------------------------------------------------------- gemv_small.hip
// One launch per Linear for small decode batches (M <= 32 rows): LayerNorm of the input rows inside the kernel, W read once,
// K slices combined inside the workgroup, bias / GELU / residual in the epilogue.
struct GemvSmallParams {
    const h16* A; int lda; int M; int K;     // input rows: the residual stream x (with ln_g: LayerNorm'ed by the kernel) or ctx / hid
    const void* Wt; int n_blocks; int w8;    // tile-linear weights, n_blocks = padded N / 16; w8: 0 fp16, 1 int8, 4 packed int4
    const h16* scale;                        // [n_blocks*16] (weight-only)
    int ksplit;                              // K slices = waves per workgroup, 1..16
    const h16* ln_g; const h16* ln_b;        // non-null: LayerNorm of the input rows over K channels, eps 1e-5
    int mode;                                // 0: fp32 sums -> out32 (attention kernels add bias, round)   1: fp16(gelu(fp16(y + bias))) -> out16
                                             // 2: x = fp16(x + fp16(y + bias)) in place   3: fp16(y) -> out16, col < n_valid
    const h16* bias; int gelu_kind;
    float* out32; int ld32;
    h16* out16; int ld16; int n_valid;
    h16* x; int ldx;
};
constexpr int GEMV_SMALL_MAX_M = 32;
int launch_gemv_small(const GemvSmallParams& p, hipStream_t stream);

// ---------------------------------------------------------------- gemv_chain.hip
// A chain of gemv_small Linears at ONE activation row in one launch (batch 1): stage inputs / outputs between the stages travel
// as {epoch, value} granules, the first stage reads plain memory, mode-0 results and the residual row go back to plain memory.
constexpr int CHAIN_MAX_STAGES = 6;
constexpr int CHAIN_MAX_ROWS = 8;            // activation rows one launch serves (gemv_chain.hip: kernels for 1, 2, 3-4 and 5-8 rows)
constexpr int DECODE_CHAIN_DEFAULT = 2;      // one-row groups: 0 a launch per kernel, 1 one launch per decoder layer, 2 one per token step
struct ChainStage {                                           // (an engine keeps its layers' stages in DEVICE memory: a chain's
    const void* Wt; const h16* scale; const h16* bias;         // arguments stay small -- by value they were 330 bytes, and the runtime
    const h16* ln_g; const h16* ln_b;                         // staged such argument blocks with a blit per launch under graph replay)
    int K, n_blocks;                                          // tile-linear weights as GemvSmallParams; inputs; groups of 16 output channels
    int mode;                                                 // 0: fp32 sums -> out32   1: gelu -> hidden row   2: residual row += ...
    int pad_;
};
struct ChainLayerStatic { const h16* qkv_bias; const h16* cq_bias; float kv_scale; int pad_; };     // per layer, the engine's (device memory)
struct ChainLayerIo { const void* cross_kv; void* cache; };                                         // per layer, the caller's (in the workspace)
struct GemvChainParams {
    // ONE decoder layer of a one-row group (n_layers == 0): n_stages = 5 or 6 consecutive descriptors in device memory -- out, cq,
    // cout, mlp1, mlp2 [, qkv of the next layer] -- framed by the attention stages:
    //   first   the self-attention of the row (attn_self_wg_kernel's arithmetic, one head per workgroup) from the qkv sums the launch
    //           before left in self_part [3C] (cache append included, in place), its output row on gran_c for stage 0
    //   behind stage cross_at = 1 (cq)   the cross-attention over 4 key-range pieces (attn_cross_kernel<1>'s arithmetic, a (head, piece)
    //           per workgroup, K / V rows by DMA into LDS from the start of the launch), q from gran_q, the pieces' partial results
    //           to gran_p as tagged granules [H][66][4]
    //   stage merge_at = 2 (cout)   merges them (attn_cross_combine_kernel's arithmetic) as its input row
    // or the WHOLE token step (n_layers > 0): st[0] = qkv of layer 0, then 6 per layer (5 for the last); per-layer pointers from the two
    // tables; the qkv sums travel as granules too (gran_s, 3 C entries).
    int n_stages; const ChainStage* st;
    float* out32;                                             // where a mode-0 stage leaves its sums (plain memory: the next launch's self-attention)
    int w8, gelu_kind;
    h16* x;                                                   // residual row [C]: read by the first stage that needs it, rewritten by every mode-2 stage
    unsigned long long* gran_x; unsigned long long* gran_h;   // granule edges: C / 2 and 4 C / 2 entries
    const h16* cross_kv; int cross_Tk, cross_heads, cross_nsplit; const h16* cross_qbias;
    unsigned long long* gran_q;                               // C entries ({epoch, fp32 bits})
    int cross_at, merge_at, merge_nsplit, merge_heads;
    unsigned long long* gran_p;
    const float* self_part; const h16* self_bias; void* self_cache; int self_cap, self_T, self_heads, self_i8; const int32_t* self_t_dev;
    float self_kv_scale; h16* self_out; unsigned long long* gran_c;     // C / 2 entries; self_out: optional plain copy [C] (tests)
    int n_layers; const ChainLayerStatic* lstat; const ChainLayerIo* lio; unsigned long long* gran_s;
    // ROWS.  rows = 1 | 2 activation rows (utterances) in one launch: x [rows][C], every granule edge [rows][...], out32 [rows][N],
    // self_part [rows][3C]; the rows' cross K/V and caches lie cross_row_bytes / self_row_bytes apart (the [B, 2, H, T, 64] buffers of
    // a batch).  A row's arithmetic does not depend on rows (the MFMA's A rows carry the activation rows alternately).  Two rows put
    // 2 x (heads + heads x pieces) attention workgroups in the launch; the Linear stages cost what they cost at one row.
    int rows; long cross_row_bytes, self_row_bytes;
    // optional list of the rows still decoding (wm_decoder_io::live_rows: count, then ascending indices): a finished row's attention stages
    // read nothing and append nothing to its cache -- they publish zeros -- so the step's bytes follow the live rows; a live row's result
    // does not depend on the list (the Linear stages carry every row regardless)
    const int32_t* live;
    unsigned* err;                                            // set non-zero when a bounded wait gives up
    const unsigned* generation; int launch_id;                // epochs: (*generation << 10) | (layer or launch_id << 3), + stage + 1 (generation: one per decoder call)
};
bool gemv_chain_supports(int C, int w8, int n_cu);
int gemv_chain_resident(int w8, int self_i8, int rows, int cross_Tk, int cross_nsplit, int n_wg, int n_cu, bool* ok, char* why, size_t why_cap);      // can the current device hold the launch's workgroups together?
int launch_occupy(int n_wg, size_t lds_bytes, long long usec, hipStream_t stream);      // diagnostic: workgroups that hold LDS and sleep
int launch_gemv_chain(const GemvChainParams& p, const ChainStage* host_stages, int n_wg, hipStream_t stream);
int launch_chain_io_table(ChainLayerIo* dst, const ChainLayerIo* host, int n, hipStream_t stream);      // fills the caller's table (small launches, arguments by value)     // host_stages: the same descriptors, for the argument checks

// ---------------------------------------------------------------- gemm_rows.hip
// The same contract (GemvSmallParams, modes 0-2, optional LayerNorm prologue; ksplit unused) for ANY number of rows: the rows
// are split over workgroups (16 or 32 rows x 64 channels each), the whole K per wave.  K <= 1536 (the input block sits in LDS).
bool gemm_rows_supports(int K, int w8);
int launch_gemm_rows(const GemvSmallParams& p, hipStream_t stream);

// ---------------------------------------------------------------- rowops.hip
struct RowFinishParams {
    // y = sum_s part[s][m][:] + bias ; y16 = fp16(y)
    const float* part; int ksplit; int M; int N; int ldp;   // ldp = padded N of the partial slabs
    long part_sstride;                                      // elements between slabs (0: M * ldp)
    const h16* bias;
    // mode 0: x = fp16(x + y16) (in place, residual stream), then xn = LayerNorm(x) * g + b
    // mode 1: h = fp16(gelu(y16))                      (MLP hidden)
    // mode 2: xn = LayerNorm(x) * g + b only           (no partials; first block / encoder)
    // mode 3: x = fp16(x + y16) only
    int mode; int gelu_kind;
    h16* x; int ldx;
    const h16* ln_g; const h16* ln_b;
    h16* out; int ldo;
    int eager;                                              // set by the launcher: few rows, go for latency (see launch_row_finish)
};
int launch_row_finish(const RowFinishParams& p, hipStream_t stream);

struct EmbedParams {
    const int32_t* tokens; int tokens_ld;    // token of row m = tokens[(m / L) * tokens_ld + m % L]
    int M;                                   // M = B*L rows
    int L;                                   // row m uses pos[m % L]
    const void* emb_tiles; int C;            // token embedding in fp16 tile-linear layout [V/16][C/32][64][8]
    const h16* pos;                          // [L][C] (already offset by the caller, decoding.py:604-608)
    h16* x; int ldx;
    int n_vocab;
    const int32_t* t_dev;                    // optional device counter: token column / position row offset
    unsigned* generation;                    // optional: incremented once per call (the one-row chain's epoch counter)
};
int launch_embed(const EmbedParams& p, hipStream_t stream);

int launch_mel_transpose_pad(const h16* mel, int B, int n_mels, int T, h16* out /*[B][T+2][n_mels]*/,
                             hipStream_t stream);
int launch_zero_pad_rows(h16* buf, int B, int Tpad, int C, hipStream_t stream);
int launch_layernorm(const h16* x, int ldx, int M, int N, const h16* g, const h16* b, h16* out, int ldo,
                     hipStream_t stream);

// ---------------------------------------------------------------- attn_encoder.hip
struct AttnEncParams {
    const h16* qkv; int ld;                  // [B*T][3C] rows, q | k | v column blocks (q,k pre-scaled by d^-0.25)
    int B, T, H;                             // head size 64
    h16* out; int ldo;                       // [B*T][C]
    int max_wgs;                             // > 0: persistent launch of at most that many workgroups (shared encoder)
};
int launch_attn_encoder(const AttnEncParams& p, hipStream_t stream);

// ---------------------------------------------------------------- attn_decode.hip
struct AttnSelfParams {
    const float* part; int ksplit; int ldp;  // qkv partial slabs [ksplit][M][ldp], columns q | k | v
    long part_sstride;                       // elements between slabs (0: M * ldp)
    const h16* bias;                         // [3C] = [q bias, 0, v bias] (weight.py:209-215)
    int B, L, T, H;                          // T cached tokens before this call
    const void* past; long past_bstride; int past_cap;      // [B][2][H][past_cap][64]
    void* present; long present_bstride; int present_cap;   // may alias past (in-place append)
    int int8_kv; float kv_scale;             // t (kv_quant_orig); 1/t formed in fp32
    float* amax;                             // optional: running max |q|,|k|,|v| (int8-KV calibration)
    const int32_t* t_dev;                    // optional device copy of T (overrides T; hipGraph replay)
    h16* out; int ldo;                       // [M][C]
    const int32_t* live;                     // optional [1 + B]: count, then the rows to process (others are skipped)
    int waves;                               // waves per (b, h): 0 / 1 the one-wave kernel, 4 the workgroup form (small groups)
};
int launch_attn_self(const AttnSelfParams& p, hipStream_t stream);

struct AttnCrossParams {
    const float* part; int ksplit; int ldp;  // q partial slabs [ksplit][M][ldp]
    long part_sstride;
    const h16* bias;                         // [C]
    int B, L, H, Tk;                         // Tk = n_audio_ctx (1500)
    const h16* kv; long kv_bstride;          // [B][2][H][Tk][64] fp16 (or int8 codes when kv_q8_scale > 0); stride in elements
    float kv_q8_scale;                       // > 0: K/V are int8, value = fp16(code) * scale (opt-in int8 cross K/V)
    h16* out; int ldo;                       // [M][C]
    int nsplit;                              // key-range splits per (b,h)  (1 = single pass)
    float* ws;                               // [B*H*nsplit][L][66] partial (m, l, o[64]) when nsplit > 1
    const int32_t* live;                     // optional [1 + B]: count, then the rows to process (others are skipped)
    int skip_zero_rows;                      // fp16 K/V, nsplit == 1: V rows whose probabilities all round to fp16 zero are not fetched (exact)
};
constexpr int CROSS_V_SKIP_DEFAULT = 1;      // measured: profiles/r4e_* (diffuse attention: no slower; peaked: FETCH_SIZE falls with the skipped rows)
int launch_attn_cross(const AttnCrossParams& p, hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---------------------------------------------------------------- greedy.hip
struct GreedyParams {
    h16* logits; long ld_row;                // row b at logits + b*ld_row; suppressed entries are overwritten with -inf
    int B; int V;
    int32_t* tokens; int ld_tok; int cur_len;     // [B][ld_tok], cur_len tokens valid; next written at cur_len
    float* sum_logprobs;                     // [B]
    const int32_t* suppress; int n_suppress; // token ids suppressed on every step (SuppressTokens + no_timestamps)
    const int32_t* blank; int n_blank;       // SuppressBlank list (incl. eot)
    int sample_begin; int eot; int timestamp_begin; int max_initial_ts;   // -1 = no limit
    int apply_rules;                         // 0 = plain argmax (tests / models without special ids), 1 = all rules, 2 = suppress lists only (without_timestamps)
    int32_t* n_done;                         // [1] number of rows whose last token is eot after this step
    const int32_t* t_dev;                    // optional device step counter: cur_len = *t_dev + 1
    int32_t* done;                           // optional [B]: 1 once the row's newest token is eot
    const int32_t* row_limit;                // optional [B]: at most that many sampled tokens per row, then eot (no log-prob)
    float temperature;                       // > 0: draw from softmax(logits / temperature) (Gumbel-max) instead of the arg-max
    uint32_t seed_lo, seed_hi; const uint32_t* seed_dev;     // the generator's seed: immediate, or [2] words in device memory (graph replay)
    int row0;                                // global index of row 0 (draws are keyed on the global row, not on the launch)
};
int launch_greedy(const GreedyParams& p, hipStream_t stream);
int launch_step_advance(int32_t* counter, hipStream_t stream);
int launch_step_finish(int32_t* counter, const int32_t* done, int B, int32_t* live, hipStream_t stream);
int launch_argmax(const h16* logits, long ld_row, int B, int V, int32_t* ids, hipStream_t stream);

// ---------------------------------------------------------------- frontend.hip
size_t log_mel_workspace_bytes(int batch, int n_samples, int n_mels);
int launch_log_mel(const float* audio, int batch, int n_samples, long audio_ld, const float* filters, int n_mels,
                   h16* out16, float* out32, void* workspace, size_t workspace_bytes, hipStream_t stream);

int launch_quantize_i8(const h16* x, int8_t* q, long n, float inv_scale, hipStream_t stream);

}  // namespace wm
