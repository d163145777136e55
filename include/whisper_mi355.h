/* whisper_mi355.h -- C ABI of libwhisper_mi355.so, the MI355X (gfx950) Whisper engine.
 *
 * This library is the drop-in for what sits under the reference's `Session` object
 * (R/tensorrt_llm/runtime/session.py:53-61,116-178; R = tensorrt_llm_july-release-v1): a
 * deserialised TensorRT engine plus the plugin library loaded by
 * `ctypes.CDLL(libnvinfer_plugin_tensorrt_llm.so); initLibNvInferPlugins(...)`
 * (R/tensorrt_llm/plugin/plugin.py:10-22, R/cpp/tensorrt_llm/plugins/api/InferPlugin.cpp:147-170).
 * TensorRT's IPluginV2DynamicExt vtable ABI is not reproducible without TensorRT, so the boundary
 * sits one level up: the three engines of examples/whisper (encoder, cross-attention K/V,
 * decoder; W/build.py:26-31) as three stateless entry points, with the same contract `Session.run`
 * has: asynchronous enqueue on the caller's stream, the CALLER owns every buffer (inputs, outputs
 * and workspace are raw device pointers), no hidden allocation, no hidden synchronisation.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; `wm_last_error()` then returns a
 *     thread-local message.  Nothing throws, nothing aborts (the reference's plugins are
 *     `noexcept` and report through caughtError, weightOnlyQuantMatmulPlugin.cpp:332-342).
 *   - all device pointers must belong to the device the engine was created on.
 *   - fp16 = IEEE binary16.  Tensors are dense row-major unless a leading dimension is given.
 *   - re-entrant for distinct (engine, stream, workspace); one host thread per GPU in the
 *     data-parallel design.  The only global state is the error string and WM_SYNC_CHECK.
 *     A caller that CAPTURES calls of this library into a hipGraph while another of its threads is inside the HIP runtime (the Python host
 *     issues the next batch's encoder from a helper thread) must capture with hipStreamCaptureModeThreadLocal or keep that thread out of
 *     the runtime for the length of the capture: in the global mode any thread's runtime call invalidates the capture (the Python host
 *     does both: decoding.py, encoding.py, native.CAPTURE_LOCK).
 */
#ifndef WHISPER_MI355_H
#define WHISPER_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wm_engine wm_engine;
typedef void* wm_stream_t; /* hipStream_t */

enum { WM_ENGINE_ENCODER = 0, WM_ENGINE_DECODER = 1, WM_ENGINE_CROSS_KV = 2 };
enum { WM_FLAG_WEIGHT_ONLY_INT8 = 1, WM_FLAG_INT8_KV = 2, WM_FLAG_GELU_TANH = 4,
       WM_FLAG_INT8_CROSS_KV = 16 /* opt-in, beyond the reference: cross-attention K/V stored as int8 codes */ };

/* Same ten fields, same order, as the OpenAI checkpoint `dims` (W/build.py:146-154). */
typedef struct wm_dims {
    int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} wm_dims;

/* ABI version of this header.  Callers built against another version must refuse to run (native.py does).
 *   2  wm_gemm takes a workspace before the stream; wm_decoder_io / wm_greedy_io carry the per-row `done` flags
 *   3  weight-only engines keep int8 at rest: wm_cross_kv takes (workspace, workspace_bytes) and REJECTS fewer than
 *      wm_cross_kv_workspace_bytes() bytes for every engine kind (256 B for fp16 engines; the fp16 expansion of one layer's
 *      matrix for weight-only ones) -- a caller that passed NULL / 0 to an fp16 engine before now gets rc 1;
 *      wm_gemm_rows, wm_set_rows_path
 *   4  wm_set_self_attn_waves
 *   5  (round 4) wm_set_gemm_small_tiles, wm_lab_knobs; environment knobs are honoured only under WM_LAB=1;
 *      wm_greedy_io gains the sampling fields (temperature, seed) and `without_timestamps` moves into the device rules
 *   6  (round 4) wm_set_decode_chain / wm_decode_chain_error; wm_decoder_io gains `workspace_id` (appended)
 *   7  (round 5) the decoder workspace needs NO initialisation by the caller any more (the library clears the state it keeps there,
 *      on the stream of the call); a one-launch step that gave up makes the NEXT wm_decoder_step fail (rc 1) until
 *      wm_decode_chain_error has been called; wm_decode_chain_status, wm_debug_occupy
 *   8  (round 6) wm_decoder_io gains `not_alone` (appended): the caller says when other decoder steps may run beside this one          */
#define WM_ABI_VERSION 8
int wm_version(void);
const char* wm_last_error(void);
int wm_device_count(int* out);

/* ---- engines ------------------------------------------------------------------------------
 * `blob` is the content of one *.engine file written by build.py (our packed-weight format, see
 * DESIGN.md "engine blob").  Replaces Session.from_serialized_engine (session.py:53-61).
 * The library copies the weights to `device` and keeps only that copy. */
int wm_engine_create(const void* blob, size_t nbytes, int device, wm_engine** out);
void wm_engine_destroy(wm_engine* e);
int wm_engine_info(const wm_engine* e, int32_t* kind, uint32_t* flags, wm_dims* dims);
/* bytes of device memory holding this engine's weights.  Weight-only engines keep int8 (or int4 codes, one per byte for the
 * row-major matrices of the encoder / cross-K/V engines) + fp16 scales at rest; the encoder / cross-K/V engines expand one matrix
 * at a time to fp16(fp16(q) * scale) into the caller's workspace right before its GEMM (the workspace sizes below include that
 * scratch block): the reference dequantises in registers (fpA_intB_gemm_template.h:47-140), same values.                    */
size_t wm_engine_weight_bytes(const wm_engine* e);

/* ---- encoder engine: W/encoding.py:48-76 -> WhisperEncoder.forward (whisper/model.py:149-172) ---
 * mel  : fp16 [batch, n_mels, 2*n_audio_ctx]    ("x")
 * out  : fp16 [batch, n_audio_ctx, n_audio_state] ("output")                                      */
size_t wm_encoder_workspace_bytes(const wm_engine* e, int batch);
int wm_encoder_forward(const wm_engine* e, const void* mel, int batch, void* out,
                       void* workspace, size_t workspace_bytes, wm_stream_t stream);
/* The same forward pass, sharing the GPU with other work (the reference's run.py takes its batches one after the other,
 * W/run.py:109-169; here the encoder of batch n+1 can run UNDER the HBM-bound decode loop of batch n).
 * cu_budget > 0: the MFMA-bound kernels of this call occupy at most that many compute units -- their workgroups are
 * persistent and stay on their CU, so kernels of other streams are never queued behind them; the results are bit-identical
 * to wm_encoder_forward's (same tiles, same arithmetic, fewer workgroups walking over them).  0 = the whole chip. */
int wm_encoder_forward_shared(const wm_engine* e, const void* mel, int batch, void* out,
                              void* workspace, size_t workspace_bytes, int cu_budget, wm_stream_t stream);
/* The same pass in pieces (ABI 7): layers [layer_begin, layer_end) of the encoder on `cu_budget` CUs (0 = the whole chip).  A range
 * that starts at 0 runs the two convolutions first, one that ends at n_audio_layer the final LayerNorm, which writes `out`; between
 * the calls of one pass the residual stream lives in `workspace` (same workspace, same batch, calls in layer order on one stream
 * or ordered streams).  A caller that runs the pass beside a decode loop gives back the CU budget the moment the loop has ended:
 * the layers still to come are issued with cu_budget 0 (WhisperEncoding.prefetch does, layer by layer).  Bit-identical to
 * wm_encoder_forward whatever the cut and the budgets.                                                                        */
int wm_encoder_forward_range(const wm_engine* e, const void* mel, int batch, void* out, void* workspace, size_t workspace_bytes,
                             int cu_budget, int layer_begin, int layer_end, wm_stream_t stream);

/* ---- cross-attention K/V engine: W/decoding.py:515-541 -> CrossAttn_KV.forward (model.py:469-540)
 * xa          : fp16 [batch, n_audio_ctx, n_text_state]
 * out_layers  : n_text_layer device pointers, each fp16 [batch, 2, n_head, n_audio_ctx, 64]
 *               ("cross_present_key_value_{i}", dim 1: 0 = K, 1 = V)                              */
size_t wm_cross_kv_workspace_bytes(const wm_engine* e, int batch);
int wm_cross_kv(const wm_engine* e, const void* xa, int batch, void* const* out_layers,
                void* workspace, size_t workspace_bytes, wm_stream_t stream);

/* ---- decoder engine: W/decoding.py:543-659 -> WhisperDecoder.forward (model.py:241-299) ---------
 * One call = L new tokens for each of `batch` utterances on top of T cached tokens.               */
typedef struct wm_decoder_io {
    int32_t batch, n_new /* L: 1 for a decode step, len(sot_sequence) for the prefill; blocks longer than 4
                            (prompts, prefixes) run as 4-token passes over the growing cache */, n_past /* T */;
    const int32_t* tokens;            /* int32 [batch, L]                               "x" */
    int32_t tokens_ld;                /* elements between utterances in `tokens` (0 = L); lets a
                                         step read column cur-1 of a [batch, capacity] buffer   */
    const void* positional_embedding; /* fp16 [L, n_text_state], rows T..T+L of the table
                                         (the caller slices: decoding.py:604-608)            */
    /* self-attention KV cache, per layer [batch, 2, n_head, capacity, 64], int8 when the engine
     * has WM_FLAG_INT8_KV, else fp16.  past[i] may be NULL when n_past == 0.
     * present[i] == past[i] with equal capacities = in-place append (the fast path);
     * otherwise the T cached rows are copied and the L new ones appended (the reference's
     * concat semantics, attention.py:296-306, "present_key_value_{i}" of shape [.., T+L, 64]). */
    const void* const* past;  int32_t past_capacity;
    void* const* present;     int32_t present_capacity;
    const void* const* cross; /* per layer fp16 [batch, 2, n_head, n_audio_ctx, 64]          */
    void* logits;             /* fp16 [batch, L, n_vocab]                            "output" */
    void* workspace; size_t workspace_bytes;
    /* optional calibration hook (NULL in production): fp32 [n_text_layer], running
     * max(|q|,|k|,|v|) of each layer's self-attention projections, the statistic the reference's
     * int8-KV calibration collects with forward hooks (W/smoothquant.py:117-175,
     * W/torch_whisper_convert.py:145-167).  The caller zeroes it before the first call. */
    float* qkv_amax;
    /* optional device-resident step counter (NULL = use n_past): int32 holding T.  With it, one
     * captured hipGraph of a decode step can be replayed for every token: `tokens` is then the base
     * of the [batch, tokens_ld] buffer (column T is read), `positional_embedding` the base of the
     * table (row T is read), `n_past` only an upper bound for validation.  Requires n_new == 1 and
     * in-place KV append.  wm_step_advance increments the counter on the stream.                   */
    const int32_t* n_past_dev;
    /* optional list of the utterances that are still decoding (NULL = all of them): device int32 [1 + batch], word 0 =
     * their number n, words 1..n = their indices in ascending order (wm_step_finish maintains it from the `done` flags of
     * wm_greedy_step).  The reference stops its (single) utterance at EOT (W/decoding.py:817-819); in a batch the rows that
     * have finished drop out of the attention kernels this way: their cross K/V (245.76 MB per token at large-v2) and
     * their KV cache are no longer read, their rows of `logits` are left unspecified (finite) -- wm_greedy_step keeps such a
     * row at EOT whatever its logits.  A live row's result does not depend on which other rows are live. */
    const int32_t* live_rows;
    /* identity of the workspace's CONTENTS (ABI 6): 0 = unknown.  The one-launch token step of a one-row group (wm_set_decode_chain)
     * keeps state inside the workspace: tagged granules, a call counter, a table of the per-layer `cross` / `present` pointers.  The
     * workspace itself needs no preparation (ABI 7: hipMalloc / torch.empty memory is fine) -- the library initialises that state, on
     * the stream of the call: with 0 on EVERY call (a 120 KB memset node and four small launches: correct, a few microseconds slower),
     * with a non-zero id once per (address, id), and the table is rewritten only when the pointers differ from what the library last
     * wrote there.  A non-zero id is the caller's promise that since the id was first passed nothing but wm_decoder_step calls of this
     * engine with the same (batch, n_new) have written to the workspace: give every allocation a new id, and a new one whenever the
     * memory is reused for anything else.  A call that is being captured into a graph is never remembered: if no eager call has
     * initialised the state under this (address, id) before, the graph carries the initialisation and every replay repeats it --
     * issue one eager call first (as WhisperDecoding.main_loop does) to keep it out of the graph. */
    uint64_t workspace_id;
    /* (ABI 8) non-zero: steps of OTHER utterance groups may be in flight on this device while this one runs (stream-parallel groups, as
     * WhisperDecoding.main_loop issues them from 13 utterances up).  The one-launch forms of a group of up to eight rows need their 256
     * workgroups resident TOGETHER; two such launches dispatched side by side can each hold half of the chip and wait for the other
     * half until the bounded waits give up -- so a step that is not alone always takes a launch per kernel.  0 = the caller issues one
     * decoder step at a time on this device (the reference's own schedule, W/decoding.py:785-821).  */
    int32_t not_alone;
} wm_decoder_io;
size_t wm_decoder_workspace_bytes(const wm_engine* e, int batch, int n_new);
int wm_decoder_step(const wm_engine* e, const wm_decoder_io* io, wm_stream_t stream);

/* The same step for n_groups (<= 8) independent utterance groups at once, scheduled for the chip:
 * group g's latency-bound kernels (weight-streaming GEMMs, row kernels, self-attention) are enqueued on
 * light_streams[g]; every group's cross-attention kernel -- the HBM-bound part, 245.76 MB per utterance --
 * goes to heavy_stream in (layer, group) order, tied to its group's stream by events.  With the light
 * streams confined to a few CUs and the heavy stream to the rest (wm_stream_create_cu_mask) the short
 * kernels of one group run at full speed WHILE another group streams its K/V; on ordinary streams the
 * hardware queues starve them (scripts/cumask_probe.py).  Results are identical to n_groups calls of
 * wm_decoder_step.  Eager launches only (CU masks do not survive hipGraph capture). */
int wm_decoder_step_multi(const wm_engine* e, int n_groups, const wm_decoder_io* const* ios,
                          const wm_stream_t* light_streams, wm_stream_t heavy_stream);
/* A stream whose kernels only run on the CUs whose bit is set in mask[0..n_words) (bit i of word i/32 =
 * CU i; MI355X: 256 CUs = 8 words).  hipExtStreamCreateWithCUMask behind the C ABI. */
int wm_stream_create_cu_mask(const uint32_t* mask, int n_words, wm_stream_t* out);
int wm_stream_destroy(wm_stream_t stream);

/* ---- fused greedy step (device-side restatement of W/decoding.py:134-217,274-300) ----------------
 * Applies SuppressBlank / SuppressTokens / ApplyTimestampRules to the last-position logits of each
 * utterance, picks the arg-max, accumulates its log-probability, keeps finished rows at EOT and
 * appends the token at tokens[b][cur_len].  n_done (device int32, caller zeroes it) counts rows
 * whose new token is EOT.                                                                          */
typedef struct wm_greedy_io {
    void* logits; int64_t row_stride; /* fp16; row b starts at logits + b*row_stride elements;
                                         suppressed entries are overwritten with -inf in place,
                                         like the reference's filters do (decoding.py:202-217) */
    int32_t batch, n_vocab;
    int32_t* tokens; int32_t tokens_ld; int32_t cur_len;
    float* sum_logprobs;
    const int32_t* suppress; int32_t n_suppress; /* token ids suppressed on every step */
    const int32_t* blank; int32_t n_blank; /* suppressed on the first sampled step only */
    int32_t sample_begin, eot, timestamp_begin, max_initial_timestamp_index /* -1: none */;
    int32_t apply_rules; /* 0: plain arg-max; 1: SuppressBlank + SuppressTokens + ApplyTimestampRules; 2: the two suppress filters only (without_timestamps, W/decoding.py:337-346) */
    int32_t* n_done;
    const int32_t* n_past_dev; /* optional device step counter: cur_len = *n_past_dev + 1 */
    int32_t* done;             /* optional int32 [batch]: set to 1 when the row's new token is EOT (never cleared here) */
    const int32_t* row_limit;  /* optional int32 [batch]: row b samples at most row_limit[b] tokens, then ends with EOT
                                  without adding to its log-probability -- a per-utterance `sample_len` (the reference has
                                  one for the whole loop, W/decoding.py:328, whose loop ends the same way: no EOT
                                  log-probability, finalize pads the EOT) */
    /* ABI 5: sampling (GreedyDecoder.update at temperature > 0, W/decoding.py:282-285).  temperature > 0: the next token is
     * drawn from softmax(filtered logits / temperature) by a Gumbel-max with a counter-based generator keyed on (seed, row0 + b,
     * position, token); sum_logprobs books log_softmax(filtered logits)[token] at temperature 1, as the reference does.
     * seed_dev (optional, two uint32 words in device memory: low, high) overrides `seed` -- a replayed graph then draws afresh. */
    float temperature; int32_t row0;
    uint64_t seed; const uint32_t* seed_dev;
} wm_greedy_io;
int wm_greedy_step(const wm_greedy_io* io, wm_stream_t stream);
int wm_step_advance(int32_t* counter, wm_stream_t stream);
/* End of a decode step of one utterance group: *counter += 1 (when counter is given) and the list of rows still decoding
 * is rebuilt from the flags: live[0] = number of rows with done[b] == 0, live[1..] = their indices, ascending.
 * batch <= 1024. */
int wm_step_finish(int32_t* counter, const int32_t* done, int batch, int32_t* live, wm_stream_t stream);

/* ---- kernel-level entry points (parity tests, micro-benchmarks, roofline measurement) -----------*/
/* C[M,N] = act(A[M,K] x W[N,K]^T * scale + bias) (+ residual); W fp16 or int8 (w8) row-major [N][K].
 * act: 0 none, 1 erf-GELU, 2 tanh-GELU.  Replaces CutlassFpAIntBGemmRunner::gemm /
 * the TRT MatMul (fpA_intB_gemm_template.h:47-140).  w8: the engines' own path for a weight-only
 * matrix of an M >> 16 stage -- W is expanded to fp16(fp16(q) * scale) (the reference kernels'
 * per-element dequantisation) into `workspace` (>= N*K*2 bytes), then the fp16 MFMA GEMM runs on it.
 * N a multiple of 128, K a multiple of 64.                                                         */
int wm_gemm(const void* A, int lda, int M, int K, const void* W, int N, int w8, const void* scale,
            const void* bias, const void* residual, int ldr, int act, void* C, int ldc,
            void* workspace, size_t workspace_bytes, wm_stream_t stream);
/* 1-D convolution, kernel 3, padding 1, stride 1 or 2, + GELU (gelu: 1 erf, 2 tanh), as the encoder
 * engine runs it (Conv1d, R/tensorrt_llm/layers/conv.py:52-94; W/torch_model.py:152-156): a GEMM over a
 * strided view of the zero-padded token-major input, no im2col.  x_pad fp16 [B][T_in+2][C_in] with rows 0
 * and T_in+1 of every utterance zero, followed by >= 512 finite elements of slack (the K padding is read
 * and multiplied by zero weights); W fp16 [C_out][K], K index = tap*C_in + c_in, zero columns up to a
 * multiple of 64 (weight.py: conv_weight_as_gemm); C_out a multiple of 128, C_in of 8.
 * out fp16 [B][T_in/stride][C_out].                                                                 */
int wm_conv1d_gelu(const void* x_pad, int B, int T_in, int C_in, const void* W, int K, const void* bias,
                   int C_out, int stride, int gelu, void* out, wm_stream_t stream);
/* ids[b] = arg-max of row b of fp16 logits (first index wins ties).  The decode loop itself uses
 * wm_greedy_step, which fuses this with Whisper's logit rules (apply_rules = 0: plain arg-max + append). */
int wm_argmax(const void* logits, int64_t row_stride, int batch, int n_vocab, int32_t* ids, wm_stream_t stream);
/* Weight-streaming GEMM, M <= 256, W in tile-linear layout (weight.py: tile_linear*).  w8: 0 = fp16
 * tiles, 1 = int8 tiles, 4 = packed int4 tiles (tile_linear_int4; K a multiple of 128).  `part` must
 * hold ksplit*M*n_blocks*16 floats; result[m][n] = sum_s part[s][m][n].  Replaces
 * weight_only_gemv_launcher (weightOnlyMatrixVectorMultiplication.cu:136-277,371-378).             */
int wm_gemm_skinny(const void* A, int lda, int M, int K, const void* Wt, int n_blocks, int w8,
                   const void* scale, int ksplit, float* part, wm_stream_t stream);
int wm_gemm_skinny_default_ksplit(int M, int K, int n_blocks, int w8);
/* One Linear of a SMALL decode batch (m <= 32 rows) in one launch -- what the decoder engine runs per projection at the
 * reference's own batch size of 1 (W/run.py:43-46): [LayerNorm of the input rows over k channels, eps 1e-5, when
 * ln_gamma is given] -> a . W^T (W as for wm_gemm_skinny, read once; K split over the waves of a workgroup, combined in
 * LDS) -> epilogue by `mode`:
 *   0  out32[row*ld32 + col] = the fp32 sums, scaled, without bias (the decode attention kernels add bias and round)
 *   1  out16 = fp16(gelu(fp16(y + bias)))                 (gelu_kind 1 erf, 2 tanh)
 *   2  x = fp16(x + fp16(y + bias)) in place              (residual stream)
 *   3  out16 = fp16(y), columns < n_valid                 (logits)
 * Replaces weight_only_gemv_launcher + bias / gelu / residual / LayerNorm layers (weightOnlyMatrixVectorMultiplication.cu:
 * 136-277,371-378; quantization/layer.py:311-312; functional.py:2044-2056; normalization.py:6-30).                      */
typedef struct wm_gemv_io {
    const void* a; int32_t lda, m, k;
    const void* wt; int32_t n_blocks, w8; const void* scale;
    const void* ln_gamma; const void* ln_beta;
    int32_t mode; const void* bias; int32_t gelu_kind;
    float* out32; int32_t ld32;
    void* out16; int32_t ld16, n_valid;
    void* x; int32_t ldx;
} wm_gemv_io;
int wm_gemv_fused(const wm_gemv_io* io, wm_stream_t stream);
/* Decoder calls with batch * n_new <= rows activation rows take the fused small-batch path (wm_gemv_fused per Linear);
 * default 16 (or WM_SMALL_PATH; 8 until the end of round 3), 0 = never, at most 32.  Returns the previous value.  A row's result does not depend on
 * the batch it is in on either side of the switch (one more boundary on both sides: groups of fewer than 160 (utterance, head)
 * pairs -- 8 utterances of large-v2 -- cut the cross-attention's key range into 4 pieces and merge the partial softmaxes,
 * larger ones run the exact single pass; across that boundary a row's context agrees to fp32 summation order); across the
 * switch the two paths agree to fp32 summation order (a last-bit
 * difference of fp16 values in rare cases).  Captured graphs keep the path they were captured with.                    */
int wm_set_small_batch_rows(int rows);
/* The same Linear (same wm_gemv_io, modes 0-2, optional LayerNorm prologue) for ANY number of rows m: the rows are split over
 * workgroups (16 or 32 rows x 64 channels each, the whole k per wave, the input block in LDS by DMA), so a row's sums never
 * leave the accumulators and there are neither fp32 slabs nor a row kernel.  k <= 1536.  What decoder calls with more rows
 * than the small-batch switch run for every projection whose input is n_state wide (csrc/gemm_rows.hip); replaces the same
 * reference code as wm_gemv_fused (the small-M branch of WeightOnlyQuantMatmulPlugin::enqueue,
 * weightOnlyQuantMatmulPlugin.cpp:182-197, + the element-wise layers around it).  A row's result does not depend on m or on
 * the row's position.                                                                                                      */
int wm_gemm_rows(const wm_gemv_io* io, wm_stream_t stream);
/* Decoder calls with at least min_rows activation rows (and more than the small-batch switch) use wm_gemm_rows where it
 * applies; default 40 (WM_ROWS_MIN), 0 (or WM_ROWS_PATH=0) = never: the split-K chain (wm_gemm_skinny + row kernel) for every
 * Linear, as rounds 1-2 did.  Returns the previous value.  On either side of the switch a row's result does not depend on the
 * batch it is in; across it the two forms agree to fp32 summation order.  Captured graphs keep the path they were captured with. */
int wm_set_rows_path(int min_rows);
/* Waves per (utterance, head) of the decode self-attention (wm_attn_decode_self and the decoder step): 0 (default, or
 * WM_SELF_WAVES) = by size -- calls on the small-batch side of wm_set_small_batch_rows' switch run four waves per head (key
 * range dealt over the waves, the cache read in one round trip), larger ones one wave per head; 1 or 4 = that form for every
 * size.  Returns the previous value.  The two forms round at the same points and add the softmax sum and P.V in different fp32
 * orders; the cache they append is identical.  Captured graphs keep the form they were captured with.                       */
int wm_set_self_attn_waves(int waves);
/* The MFMA-bound GEMMs (encoder layers, convolutions, cross-K/V projection; wm_gemm, wm_conv1d_gelu) pick their tile by the
 * size of the launch: with fewer than `tiles` 256 x 256 output tiles (default 150: one to four clips of large-v2, by shape) the launch
 * runs 128 x 128 tiles, two workgroups per CU (64 x 128 when even those would leave a third of the CUs idle: one clip's
 * n_state-wide projections), else the persistent 256 x 256 kernel.  0 = never the small form, < 0 = the
 * default.  Returns the previous value.  Both forms add an output element's products in the same order and share the
 * epilogue arithmetic: the results are bit-identical, a clip's encoder output does not depend on the batch it is in.     */
int wm_set_gemm_small_tiles(int tiles);
/* Batch 1 to 8 (one new token for each of up to eight utterances; batch 1 is the reference's own operating point, W/run.py:43-46;
 * 4-bit weights: one or two utterances): the decoder's kernels run as STAGES of one
 * launch (csrc/gemv_chain.hip: the stages hand the activation row over as tagged 8-byte granules -- no fences, no flags, no
 * barriers between workgroups; a stage's weights are requested before its input is waited for).  Same arithmetic as one launch
 * per kernel, bit for bit.  Modes:
 *   0  a launch per fused Linear and per attention kernel (9 per layer)
 *   1  a decoder layer as ONE launch: self-attention (cache append included), out + residual, LayerNorm + cross-attention q,
 *      the cross-attention over four key-range pieces (K / V rows by DMA into LDS while the stages before it run), the merge of
 *      the pieces + cross-attention out + residual, LayerNorm + mlp1 + GELU, mlp2 + residual, LayerNorm + qkv of the next layer
 *   2  the launch walks over the layers itself: ONE launch per token step besides the embedding and the vocabulary
 *      projection (default).  The per-layer cross K/V and cache pointers reach it through a table in the workspace.
 *      Rows: one and two utterances keep their cross-attention K / V pieces in LDS (requested a layer ahead); three to eight keep them in
 *      registers, requested at the head of the stage, two (row, head, piece) items per workgroup.  `live_rows` is honoured: a finished
 *      row's attention stages read nothing and append nothing to its cache.  A row's result is the launch-per-kernel path's, bit for bit.
 * Both need the in-place cache (past[i] == present[i], equal capacities <= 512), fp16 cross K/V and <= 32 layers for mode 2, and
 * a step that runs alone (wm_decoder_step, or wm_decoder_step_multi with one group: the launch needs all of its workgroups
 * resident together -- one per CU: do not issue such a step on a stream whose CU mask leaves it fewer CUs than the device has,
 * use mode 0 there); a call that does not qualify takes the launch-per-kernel path.
 * < 0 = default; returns the previous value.  Captured graphs keep the form they were captured with.
 * Robustness.  The launch's workgroups wait for each other, so they must be resident together.  The library therefore takes the
 * one-launch forms only when (a) a workgroup's footprint -- static + dynamic LDS and registers, from the function's attributes -- fits a
 * CU and the grid is no larger than the device's CUs (checked once per device and kernel variant; the runtime's occupancy call is
 * reported beside it but not trusted alone: ROCm 7.2's answers 0 for a footprint that runs), (b) the stream of the call owns every CU (a stream created with a CU mask that
 * leaves it fewer -- wm_stream_create_cu_mask, hipExtStreamCreateWithCUMask -- takes the launch-per-kernel path) and (c) no earlier
 * launch on the device has given up.  Every wait inside the launch is bounded (about a second): a wave that gives up sets a word in
 * pinned host memory and the rest of the launch falls through.
 * wm_decode_chain_error: *out != 0 when a workgroup of a chain gave up a wait since the last call -- the results of that step and of
 * everything decoded from it are NOT valid (another tenant held CUs or LDS while the launch was dispatched).  The call does not
 * synchronise (the word lives in host memory): ask after waiting for the steps in question, e.g. after reading their logits.  It
 * clears the word and takes the device off the one-launch forms (wm_set_decode_chain re-arms it): decode the
 * utterance again, it now runs a launch per kernel (WhisperDecoding.main_loop / detect_language do exactly that, with one warning).
 * Until it has been called, every wm_decoder_step / wm_decoder_step_multi on that device returns 1 (wm_last_error says why) -- a
 * caller that never looks cannot go on decoding from garbage.  (Calls under stream capture are exempt: they enqueue nothing.)
 * wm_decode_chain_status: what the calling thread's device has done so far (tests, diagnostics).                               */
typedef struct wm_chain_status {
    int32_t mode;               /* wm_set_decode_chain's current value */
    int32_t declined;           /* 1: the device is off the one-launch forms (a launch gave up, or the occupancy check said no) */
    int32_t error_pending;      /* 1: a give-up nobody has acknowledged yet */
    int32_t pad_;
    int64_t launches;           /* one-launch steps / layers issued (or captured) on this device */
    int64_t declined_calls;     /* decoder calls that qualified by shape but took the launch-per-kernel path for reasons (a)-(c) */
    char reason[200];           /* why the device is declined, or why the last residency check said no ("" otherwise) */
    char footprint[200];        /* the last residency check in words: LDS and registers of a workgroup against a CU's, the runtime's own figure */
} wm_chain_status;
int wm_set_decode_chain(int on);
int wm_decode_chain_error(int* out);
int wm_decode_chain_status(wm_chain_status* out);
/* Diagnostic: `n_workgroups` one-wave workgroups that each hold `lds_bytes` of LDS and sleep for `microseconds` on `stream` (a tenant
 * that keeps CUs' LDS busy: tests/test_gpu_round5.py provokes the give-up path of the one-launch step with it).                 */
int wm_debug_occupy(int n_workgroups, size_t lds_bytes, int64_t microseconds, wm_stream_t stream);
/* Exact V-row skipping in the decode cross-attention (fp16 K/V, single-pass form): a key whose softmax probability rounds to
 * fp16 zero contributes exactly nothing to P.V, so the wave instructions whose 8 rows all weigh zero do not fetch them from
 * HBM (they re-read 8 rows the workgroup has just used).  Outputs are bit-identical with it on or off for finite V.  1 = on
 * (default), 0 = off, < 0 = the default.  Returns the previous value.  Captured graphs keep the form they were captured with. */
int wm_set_cross_v_skip(int on);
/* Lab knobs (environment variables such as WM_CROSS_NSPLIT, WM_ROWS_MIN, WM_KSPLIT_CAP: DESIGN.md, scripts/README.md) change a
 * schedule or the order of fp32 sums for A/B runs.  They are honoured ONLY when WM_LAB=1 is set as well; every knob honoured by
 * this process is logged once on stderr and listed here as "NAME=value;..." (returns the length of the full list; buf may be
 * NULL).  Without WM_LAB=1 a set knob is ignored with one warning on stderr.                                                */
int wm_lab_knobs(char* buf, size_t cap);
/* fp16 LayerNorm rows, fp32 statistics, eps 1e-5 (layernormKernels.cu:62-188). */
int wm_layernorm(const void* x, int ldx, int M, int N, const void* gamma, const void* beta,
                 void* out, int ldo, wm_stream_t stream);
/* qkv fp16 [B*T, 3*H*64] with q and k pre-multiplied by 64^-0.25; out fp16 [B*T, H*64]. */
int wm_attn_encoder(const void* qkv, int ld, int B, int T, int H, void* out, int ldo, wm_stream_t stream);
/* Decode cross-attention: q fp32 [B*L, H*64] (un-scaled, bias included), kv fp16 [B,2,H,Tk,64].
 * nsplit = 1: one workgroup per (utterance, head), exact two-pass softmax.  1 < nsplit <= 16: the key range is cut into nsplit
 * pieces (ws: >= B*H*nsplit*L*66 floats) and the partial softmaxes are merged; nsplit > 16 is an error.                      */
int wm_attn_decode_cross(const float* q, int B, int L, int H, int Tk, const void* kv, void* out,
                         int nsplit, float* ws, wm_stream_t stream);
/* the same with int8 K/V codes [B,2,H,Tk,64] and one scale (value = fp16(code) * kv_scale, rounded to fp16) */
int wm_attn_decode_cross_i8(const float* q, int B, int L, int H, int Tk, const void* kv_i8, float kv_scale,
                            void* out, int nsplit, float* ws, wm_stream_t stream);
/* Decode self-attention with append: qkv fp32 [B*L, 3*H*64] (bias included); cache [B,2,H,cap,64]. */
int wm_attn_decode_self(const float* qkv, int B, int L, int T, int H, const void* past, int past_cap,
                        void* present, int present_cap, int int8_kv, float kv_scale, void* out,
                        wm_stream_t stream);
/* q = sat_s8(rne(x * inv_scale)) (quantizeTensorPlugin / attention.py:340-348). */
int wm_quantize_i8(const void* x, void* q, int64_t n, float inv_scale, wm_stream_t stream);

/* ---- audio front end ----------------------------------------------------------------
 * log_mel_spectrogram (W/whisper_utils.py:99-146) for a batch of equally long clips already padded /
 * trimmed by the caller (pad_or_trim, W/whisper_utils.py:56-81): audio fp32 [batch][audio_ld], 16 kHz,
 * n_samples a multiple of 160; filters fp32 [n_mels][201] (the mel_filters.npz matrix,
 * W/whisper_utils.py:84-96).  Writes n_frames = n_samples / 160 frames per clip as fp16 and/or fp32
 * [batch][n_mels][n_frames] (either pointer may be NULL); the max - 8 clamp is taken per clip. */
size_t wm_log_mel_workspace_bytes(int batch, int n_samples, int n_mels);
int wm_log_mel(const float* audio, int batch, int n_samples, int64_t audio_ld, const float* filters,
               int n_mels, void* mel_f16, float* mel_f32, void* workspace, size_t workspace_bytes,
               wm_stream_t stream);

/* FLAC decoder (host code, no GPU needed): replaces the ffmpeg subprocess of load_audio
 * (W/whisper_utils.py:17-54) for the LibriSpeech .flac files.  wm_flac_decode writes interleaved
 * int32 samples [n][channels]; frame CRC-8 / CRC-16 are verified, `md5` is the STREAMINFO signature of
 * the PCM (little-endian, ceil(bits/8) bytes per sample) for the caller to check. */
typedef struct wm_flac_streaminfo {
    int32_t sample_rate;
    int32_t channels;
    int32_t bits_per_sample;
    int32_t max_block_size;
    int64_t total_samples;      /* per channel; 0 = unknown */
    uint8_t md5[16];
} wm_flac_streaminfo;
int wm_flac_info(const void* data, size_t bytes, wm_flac_streaminfo* info);
int wm_flac_decode(const void* data, size_t bytes, int32_t* pcm, int64_t capacity_samples,
                   int64_t* n_decoded);

/* ---- in-situ timing of the dominant decode kernel (cross-attention), for the roofline report ------
 * When enabled, wm_decoder_step brackets the cross-attention launch of every `layer_stride`-th layer
 * with a HIP event pair on the launch stream, up to `max_samples` pairs.  wm_profile_read waits for
 * the recorded events and returns the summed duration and the number of samples.  Process-global,
 * off by default, not thread-safe (one host thread per GPU).                                        */
int wm_profile_configure(int enabled, int layer_stride, int max_samples);
int wm_profile_read(double* total_ms, int64_t* count, int reset);
/* Diagnostic: a device-side timeline of the decode step.  `buf` = int64 [1 + 3 * capacity] device words (NULL switches it
 * off): word 0 counts entries, entry i = {group tag (the step's logits pointer), 2 * layer + (0 = before, 1 = after the
 * cross-attention launch), wall clock (100 MHz)}, written by 1-thread kernels on the step's stream (graph-capturable; each
 * costs a launch, so the timeline perturbs what it shows by a few us per layer).  scripts/timeline_probe.py draws it. */
int wm_debug_timeline(void* buf, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_MI355_H */
