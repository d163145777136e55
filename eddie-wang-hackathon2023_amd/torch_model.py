"""The PyTorch comparison path of the example (`summarize.py --test_torch`, `run.py --only_torch`; BASELINE.json configs[0]: Whisper
tiny.en on the host CPU), written for this package: a FUNCTIONAL Whisper over the OpenAI checkpoint's flat state dict.

The reference ships a module tree for this (W/torch_model.py:12-302, W = /root/reference/tensorrt_llm_july-release-v1/examples/whisper:
`Whisper(ModelDimensions)` with `.encoder`, `.decoder(tokens, xa, kv_cache=)`, `.logits`, `install_kv_cache_hooks`), which
`WhisperEncoding.torch_get_audio_features` / `WhisperDecoding.torch_detect_language` / `torch_main_loop` drive (W/encoding.py:43-46,
W/decoding.py:661-701,743-783).  This file keeps that SURFACE -- the same constructor, the same four entry points, the same
checkpoint keys (W/weight.py:50-152) -- over a different design: no layer classes, no forward hooks; the parameters stay in one
dict under their checkpoint names and two plain functions walk over the layers; the KV cache is a dict the decoder fills itself
(`install_kv_cache_hooks` hands out an empty one and no hooks).

Arithmetic contract, as the reference runs it (W/summarize.py:81-84,121: fp32-stored parameters, fp16 mel): a Linear / Conv1d casts
its parameters to the activation's dtype (W/torch_model.py:30-45), LayerNorm and the softmax compute in fp32 and cast back (:25-27,
:99-103), the logits come out in fp32 (:212-214).  Pinned on CPU against the outputs the reference's own model produced
(tests/golden/model_micro.npz, model_tiny_en_shape.npz: tests/test_torch_path_cpu.py).  Nothing here is used by the HIP path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Iterable, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


@dataclass
class ModelDimensions:
    """The ten `dims` of an OpenAI checkpoint, in their order (W/torch_model.py:12-22; W/build.py:146-154)."""
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


def sinusoid_table(n_pos: int, width: int, max_timescale: float = 10000.0) -> Tensor:
    """The encoder's fixed positional table: sin | cos halves over log-spaced timescales (W/weight.py:24-30)."""
    half = width // 2
    rate = math.log(max_timescale) / (half - 1)
    angle = torch.arange(n_pos, dtype=torch.float32)[:, None] * torch.exp(-rate * torch.arange(half, dtype=torch.float32))[None, :]
    return torch.cat([angle.sin(), angle.cos()], dim=1)


def _parameter_shapes(d: ModelDimensions) -> Dict[str, Tuple[int, ...]]:
    """Every tensor of the checkpoint by name (W/weight.py:50-152 reads exactly these)."""
    shapes: Dict[str, Tuple[int, ...]] = {
        "encoder.conv1.weight": (d.n_audio_state, d.n_mels, 3), "encoder.conv1.bias": (d.n_audio_state,),
        "encoder.conv2.weight": (d.n_audio_state, d.n_audio_state, 3), "encoder.conv2.bias": (d.n_audio_state,),
        "encoder.positional_embedding": (d.n_audio_ctx, d.n_audio_state),
        "encoder.ln_post.weight": (d.n_audio_state,), "encoder.ln_post.bias": (d.n_audio_state,),
        "decoder.token_embedding.weight": (d.n_vocab, d.n_text_state),
        "decoder.positional_embedding": (d.n_text_ctx, d.n_text_state),
        "decoder.ln.weight": (d.n_text_state,), "decoder.ln.bias": (d.n_text_state,),
    }

    def block(prefix: str, c: int, cross: bool):
        for att in (("attn", "cross_attn") if cross else ("attn",)):
            for proj in ("query", "key", "value", "out"):
                shapes[f"{prefix}.{att}.{proj}.weight"] = (c, c)
                if proj != "key":                                    # Whisper's key projection has no bias
                    shapes[f"{prefix}.{att}.{proj}.bias"] = (c,)
            shapes[f"{prefix}.{att}_ln.weight"] = (c,)
            shapes[f"{prefix}.{att}_ln.bias"] = (c,)
        shapes[f"{prefix}.mlp.0.weight"] = (4 * c, c); shapes[f"{prefix}.mlp.0.bias"] = (4 * c,)
        shapes[f"{prefix}.mlp.2.weight"] = (c, 4 * c); shapes[f"{prefix}.mlp.2.bias"] = (c,)
        shapes[f"{prefix}.mlp_ln.weight"] = (c,); shapes[f"{prefix}.mlp_ln.bias"] = (c,)

    for i in range(d.n_audio_layer):
        block(f"encoder.blocks.{i}", d.n_audio_state, cross=False)
    for i in range(d.n_text_layer):
        block(f"decoder.blocks.{i}", d.n_text_state, cross=True)
    return shapes


class Whisper:
    """`Whisper(dims)`; `.load_state_dict(checkpoint["model_state_dict"])`; `.to(device)`; then the four entry points the wrappers call:
    `.encoder(mel)`, `.decoder(tokens, audio_features, kv_cache=None)`, `.logits(tokens, audio_features)`,
    `.install_kv_cache_hooks()` -> (cache dict, hooks list)."""

    def __init__(self, dims: ModelDimensions):
        self.dims = dims
        g = torch.Generator().manual_seed(0)
        self.p: Dict[str, Tensor] = {}
        for name, shape in _parameter_shapes(dims).items():          # placeholders until load_state_dict: small random values, unit gains
            if name.endswith("_ln.weight") or name.endswith("ln_post.weight") or name == "decoder.ln.weight":
                self.p[name] = torch.ones(shape)
            elif name.endswith(".bias"):
                self.p[name] = torch.zeros(shape)
            else:
                self.p[name] = torch.randn(shape, generator=g) * 0.02
        self.p["encoder.positional_embedding"] = sinusoid_table(dims.n_audio_ctx, dims.n_audio_state)
        self._causal = torch.full((dims.n_text_ctx, dims.n_text_ctx), float("-inf")).triu_(1)

    # ---- module-like conveniences the callers use ---------------------------------------------------------------------------
    def state_dict(self) -> Dict[str, Tensor]:
        return dict(self.p)

    def load_state_dict(self, state: Dict[str, Tensor], strict: bool = True):
        want = _parameter_shapes(self.dims)
        missing = [k for k in want if k not in state and k != "encoder.positional_embedding"]
        extra = [k for k in state if k not in want and not k.endswith("mask") and "alignment_heads" not in k]
        if strict and (missing or extra):
            raise KeyError(f"state dict does not fit the dimensions: missing {missing[:4]}, unexpected {extra[:4]}")
        for k, shape in want.items():
            if k in state:
                t = state[k].detach()
                if tuple(t.shape) != shape:
                    raise ValueError(f"{k}: shape {tuple(t.shape)} in the checkpoint, {shape} from the dimensions")
                self.p[k] = t.to(self.device).float()               # the reference keeps fp32 parameters and casts per layer
        return self

    def to(self, device) -> "Whisper":
        self.p = {k: v.to(device) for k, v in self.p.items()}
        self._causal = self._causal.to(device)
        return self

    def eval(self) -> "Whisper":
        return self

    @property
    def device(self):
        return self.p["decoder.ln.weight"].device

    @property
    def is_multilingual(self) -> bool:
        return self.dims.n_vocab >= 51865

    def parameters(self) -> Iterable[Tensor]:
        return self.p.values()

    # ---- the layers, as functions of the flat dict ---------------------------------------------------------------------------
    def _linear(self, name: str, x: Tensor) -> Tensor:
        b = self.p.get(name + ".bias")
        return F.linear(x, self.p[name + ".weight"].to(x.dtype), None if b is None else b.to(x.dtype))

    def _norm(self, name: str, x: Tensor) -> Tensor:
        return F.layer_norm(x.float(), (x.shape[-1],), self.p[name + ".weight"], self.p[name + ".bias"], 1e-5).to(x.dtype)

    @staticmethod
    def _attend(q: Tensor, k: Tensor, v: Tensor, n_head: int, mask: Optional[Tensor]) -> Tensor:
        b, n_q, c = q.shape
        s = (c // n_head) ** -0.25                                   # both operands carry the fourth root of 1 / head size
        qh = (q.reshape(b, n_q, n_head, -1) * s).transpose(1, 2)
        kh = (k.reshape(b, k.shape[1], n_head, -1) * s).permute(0, 2, 3, 1)
        vh = v.reshape(b, v.shape[1], n_head, -1).transpose(1, 2)
        scores = qh @ kh
        if mask is not None:
            scores = scores + mask[:n_q, :n_q]
        w = scores.float().softmax(dim=-1).to(q.dtype)
        return (w @ vh).transpose(1, 2).reshape(b, n_q, c)

    def _mlp(self, prefix: str, x: Tensor) -> Tensor:
        return self._linear(prefix + ".mlp.2", F.gelu(self._linear(prefix + ".mlp.0", self._norm(prefix + ".mlp_ln", x))))

    # ---- entry points ----------------------------------------------------------------------------------------------------------
    def encoder(self, mel: Tensor) -> Tensor:
        """mel [B, n_mels, 2 n_audio_ctx] -> audio features [B, n_audio_ctx, n_audio_state] in mel's dtype (W/torch_model.py:152-168)."""
        d, p = self.dims, self.p
        x = F.gelu(F.conv1d(mel, p["encoder.conv1.weight"].to(mel.dtype), p["encoder.conv1.bias"].to(mel.dtype), padding=1))
        x = F.gelu(F.conv1d(x, p["encoder.conv2.weight"].to(x.dtype), p["encoder.conv2.bias"].to(x.dtype), stride=2, padding=1))
        x = x.transpose(1, 2)
        if tuple(x.shape[1:]) != tuple(p["encoder.positional_embedding"].shape):
            raise ValueError(f"audio of {mel.shape[-1]} frames, the model takes {2 * d.n_audio_ctx}")
        x = (x + p["encoder.positional_embedding"]).to(x.dtype)
        for i in range(d.n_audio_layer):
            pre = f"encoder.blocks.{i}"
            h = self._norm(pre + ".attn_ln", x)
            x = x + self._linear(pre + ".attn.out", self._attend(self._linear(pre + ".attn.query", h), self._linear(pre + ".attn.key", h),
                                                                    self._linear(pre + ".attn.value", h), d.n_audio_head, None))
            x = x + self._mlp(pre, x)
        return self._norm("encoder.ln_post", x)

    embed_audio = encoder

    def decoder(self, tokens: Tensor, audio_features: Tensor, kv_cache: Optional[dict] = None) -> Tensor:
        """tokens [B, L] (all of them on the first call, the newest one afterwards when `kv_cache` is given), audio features
        [B, n_audio_ctx, C] -> fp32 logits [B, L, n_vocab] (W/torch_model.py:191-214).  `kv_cache` (the dict of
        `install_kv_cache_hooks`) keeps every layer's self-attention keys / values, appended call by call, and the cross-attention
        keys / values of the first call."""
        d, p = self.dims, self.p
        cache = kv_cache if kv_cache is not None else {}
        offset = cache["dec.0.self.k"].shape[1] if "dec.0.self.k" in cache else 0
        x = F.embedding(tokens, p["decoder.token_embedding.weight"]) + p["decoder.positional_embedding"][offset: offset + tokens.shape[-1]]
        x = x.to(audio_features.dtype)
        for i in range(d.n_text_layer):
            pre = f"decoder.blocks.{i}"
            h = self._norm(pre + ".attn_ln", x)
            k, v = self._linear(pre + ".attn.key", h), self._linear(pre + ".attn.value", h)
            if kv_cache is not None:
                for tag, new in (("k", k), ("v", v)):
                    key = f"dec.{i}.self.{tag}"
                    cache[key] = new if key not in cache else torch.cat([cache[key], new], dim=1).detach()
                k, v = cache[f"dec.{i}.self.k"], cache[f"dec.{i}.self.v"]
            x = x + self._linear(pre + ".attn.out", self._attend(self._linear(pre + ".attn.query", h), k, v, d.n_text_head, self._causal))
            h = self._norm(pre + ".cross_attn_ln", x)
            if f"dec.{i}.cross.k" not in cache:
                ck, cv = self._linear(pre + ".cross_attn.key", audio_features), self._linear(pre + ".cross_attn.value", audio_features)
                if kv_cache is not None:
                    cache[f"dec.{i}.cross.k"], cache[f"dec.{i}.cross.v"] = ck, cv
            else:
                ck, cv = cache[f"dec.{i}.cross.k"], cache[f"dec.{i}.cross.v"]
            x = x + self._linear(pre + ".cross_attn.out", self._attend(self._linear(pre + ".cross_attn.query", h), ck, cv, d.n_text_head, None))
            x = x + self._mlp(pre, x)
        x = self._norm("decoder.ln", x)
        return (x @ p["decoder.token_embedding.weight"].to(x.dtype).transpose(0, 1)).float()

    def logits(self, tokens: Tensor, audio_features: Tensor) -> Tensor:
        return self.decoder(tokens, audio_features)

    def forward(self, mel: Tensor, tokens: Tensor) -> Tensor:
        return self.decoder(tokens, self.encoder(mel))

    __call__ = forward

    def install_kv_cache_hooks(self, cache: Optional[dict] = None):
        """The protocol of W/torch_model.py:270-302 -- `(cache, hooks)`; the caller passes `cache` to every `decoder` call and removes
        the hooks when the utterance is done -- without forward hooks: the decoder above fills the dict itself, so the list of
        removable hooks is empty.  A cache that holds an utterance's keys is not for another utterance: take a fresh one."""
        return ({} if cache is None else cache), []


def load_model(checkpoint_file: str, device="cpu") -> Whisper:
    """An OpenAI `.pt` checkpoint ({'dims': ..., 'model_state_dict': ...}, W/summarize.py:78-84) as a `Whisper` on `device`."""
    checkpoint = torch.load(checkpoint_file, map_location="cpu")
    return Whisper(ModelDimensions(**checkpoint["dims"])).load_state_dict(checkpoint["model_state_dict"]).to(device)
