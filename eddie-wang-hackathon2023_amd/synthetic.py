"""Random-init Whisper checkpoints and synthetic log-mel input in the OpenAI `.pt` layout.

There is no Whisper checkpoint on any box of this build (no network), so `build.py --synthetic`,
bench.py and smoke() use weights drawn here.  `{'dims': ..., 'model_state_dict': ...}` with fp16
tensors is exactly what the reference's build.py loads (W/build.py:146-154,394).

Numbers come from numpy Philox streams, so the same seed gives the same bits on every box.
Linear weights are N(0, (gain / sqrt(fan_in))^2), the tied embedding N(0, (logit_std/sqrt(C))^2),
LayerNorm gains 1 + U(-.1, .1): see DESIGN.md "synthetic weights" for why the customary 0.02 is
not used (greedy decoding degenerates to repeating one token, parity tests become vacuous).
tests/test_synthetic.py pins this generator to the oracle's independent copy.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

DIMS = {
    "large-v2": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=1280, n_audio_head=20, n_audio_layer=32,
                     n_vocab=51865, n_text_ctx=448, n_text_state=1280, n_text_head=20, n_text_layer=32),
    "tiny.en": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=384, n_audio_head=6, n_audio_layer=4,
                    n_vocab=51864, n_text_ctx=448, n_text_state=384, n_text_head=6, n_text_layer=4),
    "micro": dict(n_mels=80, n_audio_ctx=64, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                  n_vocab=1024, n_text_ctx=32, n_text_state=128, n_text_head=2, n_text_layer=2),
    "micro-fullvocab": dict(n_mels=80, n_audio_ctx=64, n_audio_state=128, n_audio_head=2, n_audio_layer=2,
                            n_vocab=51865, n_text_ctx=448, n_text_state=128, n_text_head=2, n_text_layer=2),
}


def _sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def synthetic_state_dict(dims: dict, seed: int = 0, gain: float = 2.0, logit_std: float = 1.5,
                         ln_jitter: float = 0.1) -> Dict[str, torch.Tensor]:
    rng = np.random.Generator(np.random.Philox(seed))
    sd: Dict[str, torch.Tensor] = {}

    def normal(*shape, s=None):
        if s is None:
            s = gain / math.sqrt(shape[-1]) if len(shape) == 2 else 0.1
        return torch.from_numpy((rng.standard_normal(shape) * s).astype(np.float16))

    def ln(prefix, n):
        sd[prefix + ".weight"] = torch.from_numpy((1.0 + rng.uniform(-ln_jitter, ln_jitter, n)).astype(np.float16))
        sd[prefix + ".bias"] = torch.from_numpy(rng.uniform(-ln_jitter, ln_jitter, n).astype(np.float16))

    def attn(prefix, n):
        sd[prefix + ".query.weight"] = normal(n, n)
        sd[prefix + ".query.bias"] = normal(n)
        sd[prefix + ".key.weight"] = normal(n, n)
        sd[prefix + ".value.weight"] = normal(n, n)
        sd[prefix + ".value.bias"] = normal(n)
        sd[prefix + ".out.weight"] = normal(n, n)
        sd[prefix + ".out.bias"] = normal(n)

    def mlp(prefix, n):
        sd[prefix + ".0.weight"] = normal(4 * n, n)
        sd[prefix + ".0.bias"] = normal(4 * n)
        sd[prefix + ".2.weight"] = normal(n, 4 * n)
        sd[prefix + ".2.bias"] = normal(n)

    na, nt, n_mels = dims["n_audio_state"], dims["n_text_state"], dims["n_mels"]
    sd["encoder.conv1.weight"] = normal(na, n_mels, 3, s=1.0 / math.sqrt(3 * n_mels))
    sd["encoder.conv1.bias"] = normal(na)
    sd["encoder.conv2.weight"] = normal(na, na, 3, s=1.0 / math.sqrt(3 * na))
    sd["encoder.conv2.bias"] = normal(na)
    sd["encoder.positional_embedding"] = _sinusoids(dims["n_audio_ctx"], na).half()
    for i in range(dims["n_audio_layer"]):
        p = f"encoder.blocks.{i}"
        ln(p + ".attn_ln", na)
        attn(p + ".attn", na)
        ln(p + ".mlp_ln", na)
        mlp(p + ".mlp", na)
    ln("encoder.ln_post", na)
    sd["decoder.token_embedding.weight"] = normal(dims["n_vocab"], nt, s=logit_std / math.sqrt(nt))
    sd["decoder.positional_embedding"] = normal(dims["n_text_ctx"], nt, s=logit_std / math.sqrt(nt))
    for i in range(dims["n_text_layer"]):
        p = f"decoder.blocks.{i}"
        ln(p + ".attn_ln", nt)
        attn(p + ".attn", nt)
        ln(p + ".cross_attn_ln", nt)
        attn(p + ".cross_attn", nt)
        ln(p + ".mlp_ln", nt)
        mlp(p + ".mlp", nt)
    ln("decoder.ln", nt)
    return sd


def synthetic_checkpoint(model: str = "large-v2", seed: int = 0) -> dict:
    dims = dict(DIMS[model]) if isinstance(model, str) else dict(model)
    return {"dims": dims, "model_state_dict": synthetic_state_dict(dims, seed)}


def synthetic_mel(batch: int, n_frames: int = 3000, n_mels: int = 80, seed: int = 1234) -> torch.Tensor:
    """N(0, 0.5) clipped to [-0.5, 1.5]: the range the reference's log-mel normalisation produces
    (W/whisper_utils.py:143-145).  fp16 [batch, n_mels, n_frames]."""
    rng = np.random.Generator(np.random.Philox(seed))
    x = np.clip(rng.standard_normal((batch, n_mels, n_frames)) * 0.5, -0.5, 1.5)
    return torch.from_numpy(x.astype(np.float16))
