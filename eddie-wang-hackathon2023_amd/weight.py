"""Checkpoint -> engine tensors: weight fusion, weight-only int8 quantisation, gfx950 layouts and
the engine blob writer.

Mirror of the reference's weight loader (W/weight.py; W = /root/reference/tensorrt_llm_july-
release-v1/examples/whisper): same entry points (`load_encoder_weight`, `load_decoder_weight`,
`load_crossattn_linear_weight`), same checkpoint keys, same fusions (qkv weight = cat[q, k, v],
qkv bias = [q bias, 0, v bias], W/weight.py:64-95,196-215), same quantised set (every Linear of
the blocks and of the cross K/V engine; not the convs, the embedding, the logits matmul or the
LayerNorms -- SURVEY F7), same int8-KV scale files (W/weight.py:236-243).

What is deliberately different:
* `symmetric_quantize` is a numpy restatement of the host op the reference calls
  (`torch.ops.fastertransformer.symmetric_quantize_last_axis_of_batched_matrix`,
  R/cpp/tensorrt_llm/kernels/cutlass_kernels/cutlass_preprocessors.cpp:616-720); the NVIDIA
  layout permutations that follow it there (`preprocess_weights_for_mixed_gemm`) are replaced by
  the gfx950 layouts below.
* the cross-attention V bias IS loaded (the reference drops it by a typo, W/weight.py:372 --
  SURVEY F3; the PyTorch path is the semantics followed).
* tensors go into our own engine blob instead of TensorRT Parameters.

gfx950 layouts
  row-major [N][K]      encoder / cross-K/V weights (M >> 16 GEMM stages tiles through LDS)
  tile-linear           decoder weights and the tied embedding: [N/16][K/KT][64 lanes][16 B],
                        KT = 64 (int8) or 32 (fp16); lane l holds channel (l & 15), inputs
                        [KT/4 * (l >> 4), +KT/4): one wave-wide 16-byte load = 1 KiB contiguous,
                        already in MFMA 16x16x32 B-operand order (csrc/gemm_skinny.hip).
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

ENGINE_ENCODER, ENGINE_DECODER, ENGINE_CROSS_KV = 0, 1, 2
FLAG_WEIGHT_ONLY_INT8, FLAG_INT8_KV, FLAG_GELU_TANH, FLAG_INT8_CROSS_KV = 1, 2, 4, 16
DIM_KEYS = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
            "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")
_DTYPE_CODE = {np.dtype(np.float16): 0, np.dtype(np.int8): 1, np.dtype(np.float32): 2, np.dtype(np.int32): 3,
               np.dtype(np.uint8): 4}          # uint8 = two biased int4 values per byte (tile_linear_int4)


def _np(t) -> np.ndarray:
    """torch tensor or array -> numpy (fp16 checkpoint tensors stay fp16)."""
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t)


def _prep_device() -> torch.device:
    """Weight preparation (quantise, tile) runs as torch ops: on the GPU when there is one (large-v2 is
    1.5e9 weights; seconds instead of minutes), else multi-threaded on the host.  Same arithmetic
    either way: IEEE fp32 divide, floor, compare -- tests/test_gpu_kernels.py checks bit equality."""
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def _t(x, device=None) -> torch.Tensor:
    if not torch.is_tensor(x):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.detach().to(device or _prep_device())


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """Encoder positional table, recomputed like W/weight.py:24-30,50 and stored fp16."""
    assert channels % 2 == 0
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1).half().numpy()


# ---------------------------------------------------------------------------------------------
# weight-only int8
# ---------------------------------------------------------------------------------------------

def _weight_bits(use_weight_only) -> int:
    """0 (fp16) / 8 / 4 from the builder's `use_weight_only` + `weight_only_precision` pair: the loaders accept
    False, True (= int8), 'int8' or 'int4'."""
    if not use_weight_only:
        return 0
    if use_weight_only is True or use_weight_only == 'int8':
        return 8
    if use_weight_only == 'int4':
        return 4
    raise ValueError(f"unknown weight-only precision {use_weight_only!r}")


def _symmetric_quantize_t(w: torch.Tensor, bits: int = 8) -> Tuple[torch.Tensor, torch.Tensor]:
    w = w.float()
    half_range = float(1 << (bits - 1))
    absmax = w.abs().amax(dim=1)
    scale = absmax * (1.0 / half_range)
    safe = torch.where(scale > 0, scale, torch.ones_like(scale))
    r = w / safe[:, None]
    q = torch.trunc(r + torch.copysign(torch.full_like(r, 0.5), r))     # C round(): half away from zero
    q = q.clamp_(-half_range, half_range - 1).to(torch.int8)
    q[scale == 0] = 0
    return q, scale.half()


def symmetric_quantize(w_out_in, bits: int = 8) -> Tuple[np.ndarray, np.ndarray]:
    """Per-output-channel symmetric int8 / int4 (cutlass_preprocessors.cpp:641-708):
    scale = absmax / 2^(bits-1) in fp32, q = clip(round_half_away(w / scale), -2^(bits-1), 2^(bits-1) - 1),
    stored scale = fp16(scale).  Input [out, in] (the reference quantises the transposed [in, out] matrix per
    column, which is the same thing).  Returns (int8 [out, in] holding the integer codes, fp16 [out])."""
    q, s = _symmetric_quantize_t(_t(w_out_in), bits)
    return q.cpu().numpy(), s.cpu().numpy()


def _tile_linear_t(w: torch.Tensor) -> torch.Tensor:
    n, k = w.shape
    if w.dtype == torch.int8:
        kt, per = 64, 16
    elif w.dtype == torch.float16:
        kt, per = 32, 8
    else:
        raise TypeError(f"tile_linear: unsupported dtype {w.dtype}")
    if k % kt:
        raise ValueError(f"tile_linear: K={k} must be a multiple of {kt}")
    npad = (n + 15) // 16 * 16
    if npad != n:
        w = torch.cat([w, torch.zeros((npad - n, k), dtype=w.dtype, device=w.device)], dim=0)
    t = w.reshape(npad // 16, 16, k // kt, 4, per)          # (nb, n, kt, g, j)
    t = t.permute(0, 2, 3, 1, 4)                            # (nb, kt, g, n, j): lane = g * 16 + n
    return t.contiguous().reshape(npad // 16, k // kt, 64, per)


_NIBBLE_OF_INPUT = [0, 4, 1, 5, 2, 6, 3, 7]      # input j of an 8-input group sits at nibble j/2 (even) or 4 + j/2 (odd)


def _tile_linear_int4_t(q: torch.Tensor) -> torch.Tensor:
    """int4 codes (int8 tensor, values -8..7) [N, K] -> packed tile-linear uint8 [N/16, K/128, 64, 16]: lane
    l = 16 g + n holds channel n, inputs 128 kt + 32 g .. +32 as 32 biased nibbles (q + 8); 32-bit word m of the
    lane is the B operand of MFMA m (inputs 8 m .. 8 m + 7 of the lane), nibble order _NIBBLE_OF_INPUT
    (csrc/gemm_skinny.hip unpacks a word with four shift-and-mask steps)."""
    n, k = q.shape
    if k % 128:
        raise ValueError(f"tile_linear_int4: K={k} must be a multiple of 128")
    npad = (n + 15) // 16 * 16
    if npad != n:
        q = torch.cat([q, torch.zeros((npad - n, k), dtype=q.dtype, device=q.device)], dim=0)
    u = (q.to(torch.int16) + 8).to(torch.uint8).reshape(npad // 16, 16, k // 128, 4, 4, 8)     # (nb, n, kt, g, m, j)
    nib = torch.empty_like(u)
    nib[..., _NIBBLE_OF_INPUT] = u                                                              # (.., nibble position)
    packed = nib[..., 0::2] | (nib[..., 1::2] << 4)                                             # (nb, n, kt, g, m, byte)
    packed = packed.permute(0, 2, 3, 1, 4, 5)                                                   # (nb, kt, g, n, m, byte)
    return packed.contiguous().reshape(npad // 16, k // 128, 64, 16)


def tile_linear_int4(q) -> np.ndarray:
    return _tile_linear_int4_t(_t(q)).cpu().numpy()


def untile_linear_int4(t: np.ndarray, n: int) -> np.ndarray:
    """Inverse of tile_linear_int4 (tests): packed tiles -> int8 codes [n, K]."""
    nb, kt = t.shape[:2]
    b = t.reshape(nb, kt, 4, 16, 4, 4)                                      # (nb, kt, g, n, m, byte)
    nib = np.empty(b.shape[:-1] + (8,), dtype=np.uint8)
    nib[..., 0::2] = b & 15
    nib[..., 1::2] = b >> 4
    u = nib[..., _NIBBLE_OF_INPUT]                                          # (nb, kt, g, n, m, j)
    q = u.astype(np.int16) - 8
    return q.transpose(0, 3, 1, 2, 4, 5).reshape(nb * 16, kt * 128)[:n].astype(np.int8)


def tile_linear(w) -> np.ndarray:
    """[N, K] int8 or fp16 -> tile-linear [N/16, K/KT, 64, 16 bytes] (N padded to 16 with zeros)."""
    return _tile_linear_t(_t(w)).cpu().numpy()


def untile_linear(t: np.ndarray, n: int) -> np.ndarray:
    """Inverse of tile_linear (tests)."""
    nb, kt, _, per = t.shape
    w = t.reshape(nb, kt, 4, 16, per).transpose(0, 3, 1, 2, 4).reshape(nb * 16, kt * 4 * per)
    return w[:n]


# ---------------------------------------------------------------------------------------------
# tensor collections per engine
# ---------------------------------------------------------------------------------------------

def _linear(out: Dict[str, np.ndarray], name: str, w, b, use_weight_only: bool, tiled: bool):
    """Add one Linear: weight `[out, in]` fp16, optional bias."""
    w = _t(w).half()
    n = w.shape[0]
    bits = _weight_bits(use_weight_only)
    if bits:
        q, s = _symmetric_quantize_t(w, bits)
        if tiled:
            npad = (n + 15) // 16 * 16
            s = torch.cat([s, torch.zeros(npad - n, dtype=torch.float16, device=s.device)])
            q = _tile_linear_int4_t(q) if bits == 4 else _tile_linear_t(q)     # row-major int4 codes stay one per byte:
                                                                               # those matrices are expanded to fp16 per use (engine.hip: big())
        out[name + (".t" if tiled else ".w")] = q.cpu().numpy()
        out[name + ".s"] = s.cpu().numpy()
    else:
        out[name + (".t" if tiled else ".w")] = (_tile_linear_t(w) if tiled else w).cpu().numpy()
    if b is not None:
        out[name + ".b"] = _np(b).astype(np.float16)


def _ln(out, name, params, key):
    out[name + ".g"] = _np(params[key + ".weight"]).astype(np.float16)
    out[name + ".b"] = _np(params[key + ".bias"]).astype(np.float16)


def _qkv(params, prefix):
    """Fused qkv weight [3C, C] and bias [q, 0, v] (W/weight.py:64-95,196-215)."""
    w = torch.cat([_t(params[prefix + ".query.weight"]), _t(params[prefix + ".key.weight"]),
                   _t(params[prefix + ".value.weight"])], dim=0)
    qb = _np(params[prefix + ".query.bias"])
    b = np.concatenate([qb, np.zeros_like(qb), _np(params[prefix + ".value.bias"])], axis=0)
    return w, b


def conv_weight_as_gemm(w: np.ndarray) -> np.ndarray:
    """Conv1d weight [C_out, C_in, 3] -> the GEMM matrix [C_out, K] the engine multiplies token-major rows with:
    K index = tap * C_in + c_in, zero columns up to a multiple of 64 (the GEMM's K tile)."""
    c_out, c_in, taps = w.shape
    g = np.asarray(w, dtype=np.float16).transpose(0, 2, 1).reshape(c_out, taps * c_in)
    kpad = (taps * c_in + 63) // 64 * 64
    return np.ascontiguousarray(np.concatenate([g, np.zeros((c_out, kpad - taps * c_in), dtype=np.float16)], axis=1))


def load_encoder_weight(model_metadata: dict, model_params: dict, n_layer: int,
                        use_weight_only: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Encoder engine tensors (W/weight.py:35-152)."""
    t: "OrderedDict[str, np.ndarray]" = OrderedDict()
    C, n_mels = model_metadata["n_audio_state"], model_metadata["n_mels"]
    # convolutions as GEMMs over token-major rows: K index = tap * C_in + c_in
    t["conv1.w"] = conv_weight_as_gemm(_np(model_params["encoder.conv1.weight"]))      # [C, n_mels, 3] -> [C, 256]
    t["conv1.b"] = _np(model_params["encoder.conv1.bias"]).astype(np.float16)
    t["conv2.w"] = conv_weight_as_gemm(_np(model_params["encoder.conv2.weight"]))      # [C, C, 3] -> [C, 3C]
    t["conv2.b"] = _np(model_params["encoder.conv2.bias"]).astype(np.float16)
    t["pos"] = sinusoids(model_metadata["n_audio_ctx"], C)
    for i in range(n_layer):
        p, o = f"encoder.blocks.{i}", f"blocks.{i}"
        _ln(t, o + ".attn_ln", model_params, p + ".attn_ln")
        w, b = _qkv(model_params, p + ".attn")
        _linear(t, o + ".qkv", w, b, use_weight_only, False)
        _linear(t, o + ".out", model_params[p + ".attn.out.weight"], model_params[p + ".attn.out.bias"],
                use_weight_only, False)
        _ln(t, o + ".mlp_ln", model_params, p + ".mlp_ln")
        _linear(t, o + ".mlp1", model_params[p + ".mlp.0.weight"], model_params[p + ".mlp.0.bias"], use_weight_only, False)
        _linear(t, o + ".mlp2", model_params[p + ".mlp.2.weight"], model_params[p + ".mlp.2.bias"], use_weight_only, False)
    _ln(t, "ln_post", model_params, "encoder.ln_post")
    return t


def read_kv_scale(quantize_dir: str, layer: int) -> float:
    """`scale_y_quant_orig` of the fused self-attention qkv output, fp32[1] (W/weight.py:236-243;
    written by torch_whisper_convert.py -kv)."""
    path = os.path.join(quantize_dir, f"model.decoder.blocks.{layer}.attn.query_key_value.scale_y_quant_orig.bin")
    if not os.path.exists(path):
        raise FileNotFoundError(f"int8 KV cache needs the calibration file {path} (run torch_whisper_convert.py -kv)")
    return float(np.fromfile(path, dtype=np.float32).reshape(1)[0])


def read_cross_kv_scale(quantize_dir: str, layer: int) -> float:
    """Scale of the int8 cross-attention K/V of one layer, fp32[1] (opt-in mode; written by torch_whisper_convert.py -kv
    next to the self-attention scales)."""
    path = os.path.join(quantize_dir, f"model.decoder.blocks.{layer}.cross_attn.key_value.scale_y_quant_orig.bin")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: int8 cross K/V needs its calibration scales (torch_whisper_convert.py -kv)")
    return float(np.fromfile(path, dtype=np.float32).reshape(1)[0])


def load_decoder_weight(model_params: dict, n_layer: int, quantize_dir: Optional[str] = None,
                        use_weight_only: bool = False, use_int8_kv_cache: bool = False,
                        use_int8_cross_kv: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Decoder engine tensors (W/weight.py:154-339).  All Linears are tile-linear; the token
    embedding is stored once (fp16 tile-linear) and serves both the gather and the logits GEMM
    (the reference stores it twice: whisper/model.py:212,231)."""
    t: "OrderedDict[str, np.ndarray]" = OrderedDict()
    t["emb.t"] = _tile_linear_t(_t(model_params["decoder.token_embedding.weight"]).half()).cpu().numpy()
    for i in range(n_layer):
        p, o = f"decoder.blocks.{i}", f"blocks.{i}"
        _ln(t, o + ".attn_ln", model_params, p + ".attn_ln")
        w, b = _qkv(model_params, p + ".attn")
        _linear(t, o + ".qkv", w, b, use_weight_only, True)
        _linear(t, o + ".out", model_params[p + ".attn.out.weight"], model_params[p + ".attn.out.bias"], use_weight_only, True)
        if use_int8_kv_cache:
            t[o + ".kv_scale"] = np.array([read_kv_scale(quantize_dir, i)], dtype=np.float32)
        if use_int8_cross_kv:
            t[o + ".cross_kv_scale"] = np.array([read_cross_kv_scale(quantize_dir, i)], dtype=np.float32)
        _ln(t, o + ".cross_ln", model_params, p + ".cross_attn_ln")
        _linear(t, o + ".cq", model_params[p + ".cross_attn.query.weight"], model_params[p + ".cross_attn.query.bias"],
                use_weight_only, True)
        _linear(t, o + ".cout", model_params[p + ".cross_attn.out.weight"], model_params[p + ".cross_attn.out.bias"],
                use_weight_only, True)
        _ln(t, o + ".mlp_ln", model_params, p + ".mlp_ln")
        _linear(t, o + ".mlp1", model_params[p + ".mlp.0.weight"], model_params[p + ".mlp.0.bias"], use_weight_only, True)
        _linear(t, o + ".mlp2", model_params[p + ".mlp.2.weight"], model_params[p + ".mlp.2.bias"], use_weight_only, True)
    _ln(t, "ln", model_params, "decoder.ln")
    return t


def load_crossattn_linear_weight(model_params: dict, n_layer: int, use_weight_only: bool = False,
                                 use_int8_cross_kv: bool = False, quantize_dir: Optional[str] = None
                                 ) -> "OrderedDict[str, np.ndarray]":
    """Cross-attention K/V engine tensors (W/weight.py:341-375): per layer one fused [2C, C]
    projection (K rows then V rows), bias [0, v bias]."""
    t: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i in range(n_layer):
        p = f"decoder.blocks.{i}.cross_attn"
        wk, wv = _t(model_params[p + ".key.weight"]), _t(model_params[p + ".value.weight"])
        vb = _np(model_params[p + ".value.bias"])
        _linear(t, f"blocks.{i}.kv", torch.cat([wk, wv], dim=0),
                np.concatenate([np.zeros_like(vb), vb], axis=0), use_weight_only, False)
        if use_int8_cross_kv:
            t[f"blocks.{i}.cross_kv_scale"] = np.array([read_cross_kv_scale(quantize_dir, i)], dtype=np.float32)
    return t


# ---------------------------------------------------------------------------------------------
# engine blob
# ---------------------------------------------------------------------------------------------

def serialize_engine_blob(kind: int, flags: int, dims: dict, tensors: Dict[str, np.ndarray]) -> bytes:
    """Pack tensors into the `*.engine` format csrc/engine.hip parses (BlobHeader, BlobTensor[],
    256-byte aligned data)."""
    hdr_fmt, ent_fmt = "<8sIIII10iQQ", "<64sII4QQQ"
    n = len(tensors)
    table_end = struct.calcsize(hdr_fmt) + n * struct.calcsize(ent_fmt)
    data_offset = (table_end + 255) // 256 * 256
    entries, chunks, off = [], [], 0
    for name, arr in tensors.items():
        arr = np.ascontiguousarray(arr)
        if arr.dtype not in _DTYPE_CODE:
            raise TypeError(f"{name}: unsupported dtype {arr.dtype}")
        if arr.ndim > 4 or len(name.encode()) > 63:
            raise ValueError(f"{name}: too many dims or name too long")
        shape = list(arr.shape) + [0] * (4 - arr.ndim)
        entries.append(struct.pack(ent_fmt, name.encode(), _DTYPE_CODE[arr.dtype], arr.ndim, *shape, off, arr.nbytes))
        pad = (-arr.nbytes) % 256
        chunks.append(arr.tobytes())
        if pad:
            chunks.append(b"\0" * pad)
        off += arr.nbytes + pad
    header = struct.pack(hdr_fmt, b"WM355ENG", 1, kind, n, flags, *[int(dims.get(k, 0)) for k in DIM_KEYS],
                         data_offset, off)
    head = header + b"".join(entries)
    return head + b"\0" * (data_offset - len(head)) + b"".join(chunks)
