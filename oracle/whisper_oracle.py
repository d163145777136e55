"""CPU oracle for the Whisper hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (eddie-wang-hackathon2023_amd/) never does; it fails loudly
when the HIP library is missing instead of falling back to anything in here.

What it restates (R/ = /root/reference/tensorrt_llm_july-release-v1, W/ = R/examples/whisper):

* the model arithmetic of the reference's PyTorch path
    W/torch_model.py:25-45   (LayerNorm in fp32, Linear/Conv cast weights to the activation dtype)
    W/torch_model.py:57-103  (MultiHeadAttention: q,k scaled by d**-0.25, softmax in fp32)
    W/torch_model.py:106-135 (pre-LN residual block, exact-erf GELU)
    W/torch_model.py:138-168 (AudioEncoder: conv1/conv2 + GELU, sinusoid PE, blocks, ln_post)
    W/torch_model.py:171-214 (TextDecoder: embedding + learned PE, causal mask, tied logits)
* the weight-only int8 rule of the TensorRT-LLM path
    R/cpp/tensorrt_llm/kernels/cutlass_kernels/cutlass_preprocessors.cpp:616-720
    R/cpp/tensorrt_llm/kernels/weightOnlyMatrixVectorMultiplication.cu:44-53,136-205
    R/tests/quantization/_utils.py:37-88 (the reference's own oracle + tolerance)
* the int8 KV-cache rule
    R/tensorrt_llm/layers/attention.py:281-348 (dequant past, concat, quantise present)
    W/weight.py:236-243 (kv_orig_quant = 1/t, kv_quant_orig = t), W/utils/convert.py:76-78,98
* the cross-attention K/V projection engine
    R/tensorrt_llm/models/whisper/model.py:469-540

Pinning: tests/test_oracle_golden.py checks this file against tests/golden/*.npz, which
oracle/gen_golden.py produced IN THE BUILD CONTAINER by importing W/torch_model.py itself
(the reference is Python; it cannot travel to the GPU box, its outputs can).

Numeric modes
-------------
`act="float32"`  everything in fp32 (tolerance accounting).
`act="float16"`  the reference's fp16 mode restated platform-independently: every op reads
                 fp16-representable inputs, accumulates in fp32 and rounds its output to fp16
                 exactly where W/torch_model.py produces an fp16 tensor.  (torch's CPU half
                 kernels are not used, so the fixture does not depend on a BLAS build.)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# dimensions
# --------------------------------------------------------------------------------------


@dataclass
class Dims:
    """Same field names as the OpenAI checkpoint's `dims` dict (W/build.py:146-154)."""
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int

    def to_dict(self) -> dict:
        return asdict(self)


LARGE_V2 = Dims(80, 1500, 1280, 20, 32, 51865, 448, 1280, 20, 32)
TINY_EN = Dims(80, 1500, 384, 6, 4, 51864, 448, 384, 6, 4)
# reduced shapes used by fixtures and GPU parity tests (head size stays 64 like every Whisper)
MICRO = Dims(80, 64, 128, 2, 2, 1024, 32, 128, 2, 2)
MICRO_FULLVOCAB = Dims(80, 64, 128, 2, 2, 51865, 448, 128, 2, 2)


# --------------------------------------------------------------------------------------
# synthetic checkpoint (there is no real one on any box: SURVEY.md section 8c)
# --------------------------------------------------------------------------------------


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Encoder positional table, recomputed not loaded (W/weight.py:24-30,50)."""
    assert channels % 2 == 0
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def synthetic_state_dict(dims: Dims, seed: int = 0, gain: float = 2.0, logit_std: float = 1.5,
                         ln_jitter: float = 0.1) -> Dict[str, torch.Tensor]:
    """OpenAI-checkpoint-shaped state dict with fp16 tensors.  Tensor number i (in creation order)
    is drawn from its own numpy PCG64 stream seeded with [seed, i]: bit-stable across boxes and
    independent of how many threads the product-side generator uses.

    Key set = the keys W/weight.py reads (SURVEY.md section 8b "Checkpoint keys consumed").
    Linear weights are N(0, (gain / sqrt(fan_in))^2) so that every block contributes O(1) to the
    residual stream (with the customary 0.02 the tied embedding dominates and greedy decoding
    collapses to repeating its input token, which would make id-parity tests vacuous); the tied
    token embedding is N(0, (logit_std / sqrt(C))^2) so that logits have a spread like a trained
    model's.  LayerNorm gains are 1 + U(-j, j) and biases U(-j, j) so that a kernel that drops
    gamma or beta cannot pass; Linear biases are N(0, 0.1^2).
    """
    sd: Dict[str, torch.Tensor] = {}
    counter = [0]

    def stream():
        rng = np.random.Generator(np.random.PCG64([seed, counter[0]]))
        counter[0] += 1
        return rng

    def normal(name, *shape, s=None):
        if s is None:      # weight [out, in]: fan-in scaling; bias [out]: 0.1
            s = gain / math.sqrt(shape[-1]) if len(shape) == 2 else 0.1
        x = stream().standard_normal(shape, dtype=np.float32) * np.float32(s)
        sd[name] = torch.from_numpy(x.astype(np.float16))

    def uniform(name, n, centre):
        x = stream().random((n,), dtype=np.float32) * np.float32(2 * ln_jitter) + np.float32(centre - ln_jitter)
        sd[name] = torch.from_numpy(x.astype(np.float16))

    def ln(prefix, n):
        uniform(prefix + ".weight", n, 1.0)
        uniform(prefix + ".bias", n, 0.0)

    def attn(prefix, n):
        normal(prefix + ".query.weight", n, n)
        normal(prefix + ".query.bias", n)
        normal(prefix + ".key.weight", n, n)
        normal(prefix + ".value.weight", n, n)
        normal(prefix + ".value.bias", n)
        normal(prefix + ".out.weight", n, n)
        normal(prefix + ".out.bias", n)

    def mlp(prefix, n):
        normal(prefix + ".0.weight", 4 * n, n)
        normal(prefix + ".0.bias", 4 * n)
        normal(prefix + ".2.weight", n, 4 * n)
        normal(prefix + ".2.bias", n)

    na, nt = dims.n_audio_state, dims.n_text_state
    # conv weights scaled so that activations entering the blocks are O(1)
    normal("encoder.conv1.weight", na, dims.n_mels, 3, s=1.0 / math.sqrt(3 * dims.n_mels))
    normal("encoder.conv1.bias", na)
    normal("encoder.conv2.weight", na, na, 3, s=1.0 / math.sqrt(3 * na))
    normal("encoder.conv2.bias", na)
    sd["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, na).half()
    counter[0] += 1
    for i in range(dims.n_audio_layer):
        p = f"encoder.blocks.{i}"
        ln(p + ".attn_ln", na)
        attn(p + ".attn", na)
        ln(p + ".mlp_ln", na)
        mlp(p + ".mlp", na)
    ln("encoder.ln_post", na)

    normal("decoder.token_embedding.weight", dims.n_vocab, nt, s=logit_std / math.sqrt(nt))
    normal("decoder.positional_embedding", dims.n_text_ctx, nt, s=logit_std / math.sqrt(nt))
    for i in range(dims.n_text_layer):
        p = f"decoder.blocks.{i}"
        ln(p + ".attn_ln", nt)
        attn(p + ".attn", nt)
        ln(p + ".cross_attn_ln", nt)
        attn(p + ".cross_attn", nt)
        ln(p + ".mlp_ln", nt)
        mlp(p + ".mlp", nt)
    ln("decoder.ln", nt)
    return sd


def synthetic_checkpoint(dims: Dims, seed: int = 0) -> dict:
    """The `{'dims': ..., 'model_state_dict': ...}` object `build.py` loads (W/build.py:394)."""
    return {"dims": dims.to_dict(), "model_state_dict": synthetic_state_dict(dims, seed)}


def synthetic_mel(batch: int, n_frames: int = 3000, n_mels: int = 80, seed: int = 1234) -> torch.Tensor:
    """Synthetic log-mel: N(0, 0.5) clipped to [-0.5, 1.5], the width-2 range the reference's
    normalisation produces (W/whisper_utils.py:143-145).  fp16, [B, n_mels, n_frames]."""
    rng = np.random.Generator(np.random.PCG64([seed, batch]))
    x = np.clip(rng.standard_normal((batch, n_mels, n_frames), dtype=np.float32) * np.float32(0.5), -0.5, 1.5)
    return torch.from_numpy(x.astype(np.float16))


# --------------------------------------------------------------------------------------
# weight-only int8 (per output channel, symmetric)
# --------------------------------------------------------------------------------------


def symmetric_quantize_int8(w_out_in: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Per-output-channel symmetric int8 quantisation.

    Follows cutlass_preprocessors.cpp:641-686 on the `[K, N]` transposed weight, i.e. per
    column of W^T = per row of the `[out, in]` Linear weight given here:
        scale_f32 = absmax / 128          (:641 quant_range_scale, :667-671)
        q = clip(round_half_away(w / scale_f32), -128, 127)   (:683-686; C `round`)
        stored scale = fp16(scale_f32)    (:671 ComputeType = half; weightOnlyQuantOp.cpp:189)
    The division uses the UNROUNDED fp32 scale, the stored scale is rounded to fp16.
    An all-zero channel would divide by zero in the reference; here it yields q = 0, scale = 0.
    Returns (q int8 [out, in], scales fp16 [out]).
    """
    w = np.asarray(w_out_in, dtype=np.float32)
    absmax = np.abs(w).max(axis=1)
    scale = (absmax * np.float32(1.0 / 128.0)).astype(np.float32)
    safe = np.where(scale > 0, scale, np.float32(1.0))
    r = w / safe[:, None]
    q = np.sign(r) * np.floor(np.abs(r) + np.float32(0.5))   # C round(): half away from zero
    q = np.clip(q, -128, 127).astype(np.int8)
    q[scale == 0] = 0
    return q, scale.astype(np.float16)


def symmetric_quantize_int4(w_out_in: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Per-output-channel symmetric int4 (`--weight_only_precision int4`, W/build.py:102-112).

    Same routine as int8 with bits_in_type = 4 (cutlass_preprocessors.cpp:641 quant_range_scale = 1 / 2^(bits-1);
    :700-703 `round`, then clamp to [-8, 7] before packing): scale_f32 = absmax / 8, stored as fp16.
    Returns (int8 array holding the codes -8..7 [out, in], scales fp16 [out])."""
    w = np.asarray(w_out_in, dtype=np.float32)
    absmax = np.abs(w).max(axis=1)
    scale = (absmax * np.float32(1.0 / 8.0)).astype(np.float32)
    safe = np.where(scale > 0, scale, np.float32(1.0))
    r = w / safe[:, None]
    q = np.sign(r) * np.floor(np.abs(r) + np.float32(0.5))   # C round(): half away from zero
    q = np.clip(q, -8, 7).astype(np.int8)
    q[scale == 0] = 0
    return q, scale.astype(np.float16)


def dequantize_int8(q: np.ndarray, scales_f16: np.ndarray) -> np.ndarray:
    """fp16(fp16(q) * scale_fp16): the per-element dequantisation both reference kernels apply
    before the multiply (weightOnlyMatrixVectorMultiplication.cu:44-53 `halves[i] *= scale`)."""
    return (q.astype(np.float16) * scales_f16[:, None]).astype(np.float16)


def woq_reference_matmul(x: np.ndarray, q_in_out: np.ndarray, scales: np.ndarray) -> np.ndarray:
    """The reference test's own ground truth (R/tests/quantization/_utils.py:37-63):
    fp32 `(x @ q) * scale`, cast to fp16.  x [M,K] fp16, q [K,N] int8, scales [N] fp16."""
    ref = x.astype(np.float32) @ q_in_out.astype(np.float32)
    return (ref * scales.astype(np.float32)[None, :]).astype(np.float16)


def woq_gemv_reference(x: np.ndarray, q_in_out: np.ndarray, scales: np.ndarray) -> np.ndarray:
    """The reference's OWN M = 1 kernel, restated (int8_weight_only_gemv_interleave,
    R/cpp/tensorrt_llm/kernels/weightOnlyMatrixVectorMultiplication.cu:136-205; the plugin takes it exactly when the activation
    has one row, weightOnlyQuantMatmulPlugin.cpp:182-197):
        w16 = fp16(fp16(q) * scale)                        (:44-53 `halves[i] *= scale`)
        p   = fp16(x * w16)              PER ELEMENT       (:187 `__hmul`: every product is rounded to fp16 ...)
        y   = fp16(sum_fp32(p))                            (:187 `v += __half2float(..)`, :190-193 shuffles, :198-203 `__float2half_rn`)
    The M > 1 path of the same plugin (CUTLASS fpA_intB, default_fpA_intB_traits.h:30-109) multiplies the same w16 on tensor cores:
    EXACT fp16 x fp16 products into an fp32 accumulator -- so the reference disagrees with itself between M = 1 and M > 1 by the
    rounding of every product.  x [K] or [1, K] fp16, q [K, N] int8, scales [N] fp16 -> [1, N] fp16.
    (The order of the fp32 additions -- 16 elements per lane and stride, then four shuffles -- is not restated: numpy's pairwise
    fp32 sum differs from it by fp32 summation order only, ~1e-7 relative, four orders below the product rounding.)"""
    x = np.asarray(x, dtype=np.float16).reshape(-1)
    w16 = (q_in_out.astype(np.float16) * np.asarray(scales, dtype=np.float16)[None, :]).astype(np.float16)      # [K, N]
    p = (x[:, None] * w16).astype(np.float16)          # numpy multiplies halves in fp32 (exact) and rounds once: = __hmul
    return p.astype(np.float32).sum(axis=0, dtype=np.float32).astype(np.float16)[None, :]


def woq_colwise_atol(ref: np.ndarray) -> np.ndarray:
    """Tolerance the reference accepts (R/tests/quantization/_utils.py:66-88):
    per column 1.5 * max(col) / 128 for M > 1, one global bound for M == 1."""
    ref = ref.astype(np.float32)
    if ref.shape[0] > 1:
        return 1.5 * ref.max(axis=0) / 128.0
    return np.full(ref.shape[1], 1.5 * ref.max() / 128.0, dtype=np.float32)


# --------------------------------------------------------------------------------------
# int8 KV
# --------------------------------------------------------------------------------------


def kv_quantize(x: torch.Tensor, t: float) -> torch.Tensor:
    """int8(clip(round_half_even(x * (1/t)), -128, 127)) (attention.py:340-348; the MMHA twin is
    cvt.rni.sat.s8.f32, decoderMaskedMultiheadAttentionUtils.h:2276-2286).  1/t is formed in
    fp32 like W/weight.py:242 does (`1.0 / t` on an fp32 numpy array)."""
    inv = np.float32(1.0) / np.float32(t)
    return torch.clamp(torch.round(x.float() * float(inv)), -128, 127).to(torch.int8)


def kv_dequantize(q: torch.Tensor, t: float, act: str) -> torch.Tensor:
    """fp16(int8) * t (attention.py:283-290)."""
    y = q.float() * float(np.float32(t))
    return _r(y, act)


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------


def _r(x: torch.Tensor, act: str) -> torch.Tensor:
    """Round to the activation dtype, keep fp32 storage."""
    return x.half().float() if act == "float16" else x


@dataclass
class OracleConfig:
    act: str = "float16"              # "float16" | "float32"
    gelu: str = "erf"                 # "erf" (torch path) | "tanh" (TRT path, functional.py:2044-2056)
    weight_only: object = False       # F7: False, True (= int8) or 'int4'
    int8_kv: bool = False             # F2
    kv_scales: Optional[List[float]] = None   # t per decoder layer (F8)
    # int8 cross-attention K/V (BEYOND the reference, which keeps them fp16; an opt-in engine mode): codes =
    # sat_s8(rne(x / t)) of the fp16 projection output; attention uses the exact fp32 values code * t (no fp16 rounding
    # of the dequantised K / V: scores = r16(qh . (code_k * t * d^-0.25)), out = r16(w . (code_v * t))).
    int8_cross_kv: bool = False
    cross_kv_scales: Optional[List[float]] = None   # t per decoder layer: max(|K|, |V|) / 127 of that layer's cross K/V
    # The reference's weight-only plugin takes its GEMV kernel when the activation has ONE row, and that kernel rounds every product
    # to fp16 before the fp32 sum (woq_gemv_reference above); with more rows it takes CUTLASS: exact products.  False (default): the
    # CUTLASS contract at every M -- what the engine's Linears compute (DESIGN.md section 2 "M = 1").  True: weight-only Linears
    # whose input is one row use the GEMV kernel's arithmetic, as the reference's engines do at batch 1 after the prefill.
    gemv_fp16_products: bool = False


def _gelu(x: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "erf":
        return F.gelu(x)
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


class OracleModel:
    """Parameters prepared once (fp32 storage of fp16-representable values; optionally the
    weight-only int8 fake-quantised Linear weights), then pure functions of the inputs."""

    # Linear layers the reference quantises: every Linear of encoder blocks,
    # decoder blocks and the cross-K/V projection.  Not: convs, embedding, logits matmul.
    def __init__(self, dims: Dims, state_dict: Dict[str, torch.Tensor], cfg: OracleConfig):
        self.dims, self.cfg = dims, cfg
        self.keep_pre_quant: Optional[Dict[int, list]] = None     # tests set {}: per layer, every call's new k / v [B,2,H,L,64]
        self.p: Dict[str, torch.Tensor] = {}
        self.q: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        for k, v in state_dict.items():
            v = v.detach()
            is_linear_w = v.ndim == 2 and k.endswith(".weight") and ".blocks." in k and "_ln" not in k
            if cfg.weight_only and is_linear_w:
                quantize = symmetric_quantize_int4 if cfg.weight_only == 'int4' else symmetric_quantize_int8
                q, s = quantize(v.float().numpy())
                self.q[k] = (q, s)
                self.p[k] = torch.from_numpy(dequantize_int8(q, s).astype(np.float32))
            else:
                self.p[k] = v.float()
        if cfg.int8_kv:
            assert cfg.kv_scales is not None and len(cfg.kv_scales) == dims.n_text_layer

    def to(self, device) -> "OracleModel":
        """Move the parameters (tests run the full-size oracle on the GPU: same torch ops, fp32 matmuls, minutes
        become seconds).  Inputs must then live on that device too."""
        self.p = {k: v.to(device) for k, v in self.p.items()}
        return self

    # -- primitives --------------------------------------------------------------------
    def _linear(self, x, wkey, bkey=None):
        if self.cfg.gemv_fp16_products and wkey in self.q and x.numel() == x.shape[-1] and self.cfg.act == "float16":
            # one activation row through a weight-only Linear: the reference's GEMV kernel (fp16-rounded products, fp32 sum); the
            # bias is added to the fp16 result by a separate element-wise layer (quantization/layer.py:311-312): a second rounding
            w16 = self.p[wkey]                                        # [N, K], fp16-representable (dequantize_int8)
            prod = (x.reshape(1, -1) * w16).half().float()            # exact fp32 products of fp16 values, each rounded to fp16
            y = prod.sum(dim=1, dtype=torch.float32).half().float().reshape(*x.shape[:-1], -1)
            if bkey is not None:
                y = y + self.p[bkey]
            return _r(y, self.cfg.act)
        y = x @ self.p[wkey].t()
        if bkey is not None:
            y = y + self.p[bkey]
        return _r(y, self.cfg.act)

    def _ln(self, x, prefix):
        # W/torch_model.py:25-27: fp32 LayerNorm, eps 1e-5, cast back
        y = F.layer_norm(x, (x.shape[-1],), self.p[prefix + ".weight"], self.p[prefix + ".bias"], 1e-5)
        return _r(y, self.cfg.act)

    def _attend(self, q, k, v, n_head, mask=None, k_exact=False):
        """W/torch_model.py:88-103.  q [B,Lq,C], k/v [B,Lk,C] -> [B,Lq,C].  `k_exact`: K (already an exact product
        code * t of the opt-in int8 cross-K/V mode) is scaled in fp32 without the fp16 rounding a stored fp16 K gets."""
        act = self.cfg.act
        B, Lq, C = q.shape
        d = C // n_head
        scale = d ** -0.25
        qh = _r(q.view(B, Lq, n_head, d).permute(0, 2, 1, 3) * scale, act)
        kh = k.view(B, -1, n_head, d).permute(0, 2, 3, 1) * scale
        if not k_exact:
            kh = _r(kh, act)
        vh = v.view(B, -1, n_head, d).permute(0, 2, 1, 3)
        qk = _r(qh @ kh, act)
        if mask is not None:
            qk = qk + mask
        w = _r(torch.softmax(qk.float(), dim=-1), act)
        o = _r(w @ vh, act)
        return o.permute(0, 2, 1, 3).reshape(B, Lq, C)

    def _mlp(self, x, prefix):
        h = self._linear(x, prefix + ".0.weight", prefix + ".0.bias")
        h = _r(_gelu(h, self.cfg.gelu), self.cfg.act)
        return self._linear(h, prefix + ".2.weight", prefix + ".2.bias")

    # -- encoder (a1, a2, a3) ----------------------------------------------------------
    def encoder(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B, n_mels, 2*n_audio_ctx] -> audio features [B, n_audio_ctx, n_audio_state]."""
        act, d = self.cfg.act, self.dims
        x = _r(mel.float(), act)
        x = _r(F.conv1d(x, self.p["encoder.conv1.weight"], self.p["encoder.conv1.bias"], padding=1), act)
        x = _r(_gelu(x, self.cfg.gelu), act)
        x = _r(F.conv1d(x, self.p["encoder.conv2.weight"], self.p["encoder.conv2.bias"], stride=2, padding=1), act)
        x = _r(_gelu(x, self.cfg.gelu), act)
        x = x.permute(0, 2, 1)
        assert x.shape[1:] == (d.n_audio_ctx, d.n_audio_state), "incorrect audio shape"
        # the table reaches both reference paths rounded to fp16: the checkpoint buffer is fp16
        # (torch path) and W/weight.py:50 assigns the recomputed table to an fp16 Parameter
        pe = sinusoids(d.n_audio_ctx, d.n_audio_state).half().float().to(x.device)
        x = _r(x + pe, act)
        for i in range(d.n_audio_layer):
            p = f"encoder.blocks.{i}"
            h = self._ln(x, p + ".attn_ln")
            q = self._linear(h, p + ".attn.query.weight", p + ".attn.query.bias")
            k = self._linear(h, p + ".attn.key.weight")
            v = self._linear(h, p + ".attn.value.weight", p + ".attn.value.bias")
            a = self._attend(q, k, v, d.n_audio_head)
            x = _r(x + self._linear(a, p + ".attn.out.weight", p + ".attn.out.bias"), act)
            x = _r(x + self._mlp(self._ln(x, p + ".mlp_ln"), p + ".mlp"), act)
        return self._ln(x, "encoder.ln_post")

    # -- cross K/V engine (a5) ---------------------------------------------------------
    def cross_kv(self, xa: torch.Tensor) -> List[torch.Tensor]:
        """xa [B, n_audio_ctx, C] -> per layer [B, 2, H, n_audio_ctx, 64] (dim1: 0 = K, 1 = V).
        V carries its bias (torch semantics; the TRT path's missing bias is reference bug F3)."""
        d = self.dims
        B, T, C = xa.shape
        H = d.n_text_head
        out = []
        for i in range(d.n_text_layer):
            p = f"decoder.blocks.{i}.cross_attn"
            k = self._linear(xa, p + ".key.weight")
            v = self._linear(xa, p + ".value.weight", p + ".value.bias")
            k = k.view(B, T, H, C // H).permute(0, 2, 1, 3)
            v = v.view(B, T, H, C // H).permute(0, 2, 1, 3)
            kv = torch.stack([k, v], dim=1).contiguous()
            if self.cfg.int8_cross_kv:
                kv = kv_quantize(kv, self.cfg.cross_kv_scales[i])
            out.append(kv)
        return out

    def calibrate_cross_kv_scales(self, mels: torch.Tensor) -> List[float]:
        """t_i = max(|K_i|, |V_i|) / 127 over the cross K/V of `mels` (the rule F8 applies to the self-attention cache)."""
        saved = self.cfg.int8_cross_kv
        self.cfg.int8_cross_kv = False
        try:
            ckv = self.cross_kv(self.encoder(mels))
        finally:
            self.cfg.int8_cross_kv = saved
        return [float(np.float32(float(c.abs().max())) / np.float32(127.0)) for c in ckv]

    # -- decoder (a3, a4, a6, a11) -----------------------------------------------------
    def decoder(self, tokens: torch.Tensor, cross_kv: List[torch.Tensor],
                self_kv: Optional[List[torch.Tensor]] = None
                ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """tokens [B, L] int; self_kv per layer [B, 2, H, T, 64] (fp, or int8 when cfg.int8_kv) or None.
        Returns (logits fp32 [B, L, n_vocab], present per layer [B, 2, H, T+L, 64])."""
        act, d, cfg = self.cfg.act, self.dims, self.cfg
        B, L = tokens.shape
        H, C = d.n_text_head, d.n_text_state
        T = 0 if self_kv is None else self_kv[0].shape[3]
        emb = self.p["decoder.token_embedding.weight"]
        x = _r(emb[tokens.long()] + self.p["decoder.positional_embedding"][T:T + L], act)
        # causal mask over [past | new] keys (W/torch_model.py:186-187,209)
        mask = torch.zeros(L, T + L, device=x.device)
        mask[:, T:] = torch.full((L, L), float("-inf"), device=x.device).triu_(1)
        presents = []
        for i in range(d.n_text_layer):
            p = f"decoder.blocks.{i}"
            h = self._ln(x, p + ".attn_ln")
            q = self._linear(h, p + ".attn.query.weight", p + ".attn.query.bias")
            k = self._linear(h, p + ".attn.key.weight")
            v = self._linear(h, p + ".attn.value.weight", p + ".attn.value.bias")
            k_new = k.view(B, L, H, C // H).permute(0, 2, 1, 3)
            v_new = v.view(B, L, H, C // H).permute(0, 2, 1, 3)
            new = torch.stack([k_new, v_new], dim=1)          # [B,2,H,L,64]
            if self.keep_pre_quant is not None:               # tests: the values the int8 cache codes are rounded FROM
                self.keep_pre_quant.setdefault(i, []).append(new.clone())
            if cfg.int8_kv:
                t = cfg.kv_scales[i]
                new_q = kv_quantize(new, t)
                if self_kv is None:
                    present, full = new_q, new
                else:
                    present = torch.cat([self_kv[i], new_q], dim=3)
                    # attention sees dequantised past + full-precision current (attention.py:296-306)
                    full = torch.cat([kv_dequantize(self_kv[i], t, act), new], dim=3)
            else:
                full = new if self_kv is None else torch.cat([self_kv[i].float(), new], dim=3)
                present = full
            presents.append(present)
            k_all = full[:, 0].permute(0, 2, 1, 3).reshape(B, T + L, C)
            v_all = full[:, 1].permute(0, 2, 1, 3).reshape(B, T + L, C)
            a = self._attend(q, k_all, v_all, H, mask)
            x = _r(x + self._linear(a, p + ".attn.out.weight", p + ".attn.out.bias"), act)

            h = self._ln(x, p + ".cross_attn_ln")
            q = self._linear(h, p + ".cross_attn.query.weight", p + ".cross_attn.query.bias")
            if cfg.int8_cross_kv:
                ckv_i = cross_kv[i].float() * float(np.float32(cfg.cross_kv_scales[i]))
            else:
                ckv_i = cross_kv[i].float()
            ck = ckv_i[:, 0].permute(0, 2, 1, 3).reshape(B, -1, C)
            cv = ckv_i[:, 1].permute(0, 2, 1, 3).reshape(B, -1, C)
            a = self._attend(q, ck, cv, H, k_exact=cfg.int8_cross_kv)
            x = _r(x + self._linear(a, p + ".cross_attn.out.weight", p + ".cross_attn.out.bias"), act)

            x = _r(x + self._mlp(self._ln(x, p + ".mlp_ln"), p + ".mlp"), act)
        x = self._ln(x, "decoder.ln")
        logits = _r(x @ emb.t(), act)
        return logits, presents

    # -- int8-KV calibration (a11 / F8) -------------------------------------------------
    def calibrate_kv_scales(self, mels: torch.Tensor, n_steps: int, token_fn=None) -> List[float]:
        """t_i = max(|q|,|k|,|v| outputs of decoder self-attention layer i) / 127 over a greedy
        decode of `mels` (W/smoothquant.py:117-175, W/torch_whisper_convert.py:145-167,
        W/utils/convert.py:76-78).  Q outputs are included on purpose (reference quirk F8)."""
        d = self.dims
        amax = [0.0] * d.n_text_layer
        saved = (self.cfg.int8_kv, self.cfg.kv_scales)
        self.cfg.int8_kv, self.cfg.kv_scales = False, None
        orig_linear = self._linear

        def spy(x, wkey, bkey=None):
            y = orig_linear(x, wkey, bkey)
            if wkey.startswith("decoder.blocks.") and ".attn." in wkey and ".cross_attn." not in wkey \
                    and wkey.split(".")[4] in ("query", "key", "value"):
                i = int(wkey.split(".")[2])
                amax[i] = max(amax[i], float(y.abs().max()))
            return y

        self._linear = spy
        try:
            xa = self.encoder(mels)
            ckv = self.cross_kv(xa)
            B = mels.shape[0]
            tokens = torch.full((B, 3), 0, dtype=torch.long) if token_fn is None else token_fn(B)
            kv = None
            cur = tokens
            for _ in range(n_steps):
                logits, kv = self.decoder(cur, ckv, kv)
                cur = logits[:, -1].argmax(-1, keepdim=True)
        finally:
            self._linear = orig_linear
            self.cfg.int8_kv, self.cfg.kv_scales = saved
        return [float(np.float32(a) / np.float32(127.0)) for a in amax]


def kv_amax_on_token_path(model: OracleModel, mels: torch.Tensor, passes: List[List[torch.Tensor]]) -> List[float]:
    """The calibration statistic of `OracleModel.calibrate_kv_scales` (max |q|, |k|, |v| outputs of every decoder
    self-attention layer, W/smoothquant.py:117-175) over a GIVEN token path: `passes` is a list of decoder passes, each
    a list of token blocks [B, L] fed one after the other to a cache that starts empty (language-ID pass = [[sot]],
    main loop = [start sequence, token, token, ...]).  Teacher-forcing the path the engine decoded removes the one
    thing a calibration comparison must not depend on: which way a near-tie fell."""
    d = model.dims
    amax = [0.0] * d.n_text_layer
    saved = (model.cfg.int8_kv, model.cfg.kv_scales)
    model.cfg.int8_kv, model.cfg.kv_scales = False, None
    orig_linear = model._linear

    def spy(x, wkey, bkey=None):
        y = orig_linear(x, wkey, bkey)
        if wkey.startswith("decoder.blocks.") and ".attn." in wkey and ".cross_attn." not in wkey \
                and wkey.split(".")[4] in ("query", "key", "value"):
            i = int(wkey.split(".")[2])
            amax[i] = max(amax[i], float(y.abs().max()))
        return y

    model._linear = spy
    try:
        with torch.no_grad():
            ckv = model.cross_kv(model.encoder(mels))
            for blocks in passes:
                kv = None
                for tok in blocks:
                    _, kv = model.decoder(tok, ckv, kv)
    finally:
        model._linear = orig_linear
        model.cfg.int8_kv, model.cfg.kv_scales = saved
    return amax


def greedy_reference_run(model: OracleModel, mel: torch.Tensor, prompt: List[int], n_steps: int):
    """Plain greedy decode without Whisper's logit rules: encoder -> cross K/V -> prefill ->
    n_steps single-token steps.  Returns dict of everything a parity test compares."""
    xa = model.encoder(mel)
    ckv = model.cross_kv(xa)
    B = mel.shape[0]
    tokens = torch.tensor([prompt] * B, dtype=torch.long, device=mel.device)
    logits_all, ids, margins = [], [], []
    kv = None
    cur = tokens
    for _ in range(n_steps):
        logits, kv = model.decoder(cur, ckv, kv)
        last = logits[:, -1]
        top2 = last.topk(2, dim=-1).values
        margins.append((top2[:, 0] - top2[:, 1]).clone())
        nxt = last.argmax(-1)
        logits_all.append(logits)
        ids.append(nxt.clone())
        cur = nxt[:, None]
    return {"xa": xa, "cross_kv": ckv, "logits": logits_all, "ids": torch.stack(ids, 1),
            "margins": torch.stack(margins, 1), "self_kv": kv}
