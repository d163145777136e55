"""Random-init Whisper checkpoints and synthetic log-mel input in the OpenAI `.pt` layout.

There is no Whisper checkpoint on any box of this build (no network), so `build.py --synthetic`,
bench.py and smoke() use weights drawn here.  `{'dims': ..., 'model_state_dict': ...}` with fp16
tensors is exactly what the reference's build.py loads (W/build.py:146-154,394).

Numbers come from per-tensor numpy PCG64 streams, so the same seed gives the same bits on every box.
Linear weights are N(0, (gain / sqrt(fan_in))^2), the tied embedding N(0, (logit_std/sqrt(C))^2),
LayerNorm gains 1 + U(-.1, .1): see DESIGN.md "synthetic weights" for why the customary 0.02 is
not used (greedy decoding degenerates to repeating one token, parity tests become vacuous).
tests/test_host_cpu.py (test_synthetic_weights_match_the_oracle_generator) pins this generator to the oracle's independent copy.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

DIMS = {
    "large-v2": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=1280, n_audio_head=20, n_audio_layer=32,
                     n_vocab=51865, n_text_ctx=448, n_text_state=1280, n_text_head=20, n_text_layer=32),
    # large-v2's width and vocabulary at a depth that builds in seconds: scheduling tests and probes (queues, data parallelism)
    "large-v2-6layer": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=1280, n_audio_head=20, n_audio_layer=6,
                            n_vocab=51865, n_text_ctx=448, n_text_state=1280, n_text_head=20, n_text_layer=6),
    "tiny.en": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=384, n_audio_head=6, n_audio_layer=4,
                    n_vocab=51864, n_text_ctx=448, n_text_state=384, n_text_head=6, n_text_layer=4),
    "tiny": dict(n_mels=80, n_audio_ctx=1500, n_audio_state=384, n_audio_head=6, n_audio_layer=4,
                 n_vocab=51865, n_text_ctx=448, n_text_state=384, n_text_head=6, n_text_layer=4),
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


def _tensor_specs(dims: dict, gain: float, logit_std: float):
    """Ordered (name, shape, kind, scale) list: kind 'n' = N(0, scale^2), 'g' = 1 + U(-scale, scale),
    'u' = U(-scale, scale), 'pe' = encoder sinusoid table."""
    specs = []

    def w(name, *shape, s=None):
        if s is None:
            s = gain / math.sqrt(shape[-1]) if len(shape) == 2 else 0.1
        specs.append((name, shape, "n", s))

    def ln(prefix, n):
        specs.append((prefix + ".weight", (n,), "g", 0.1))
        specs.append((prefix + ".bias", (n,), "u", 0.1))

    def attn(prefix, n):
        w(prefix + ".query.weight", n, n); w(prefix + ".query.bias", n)
        w(prefix + ".key.weight", n, n)
        w(prefix + ".value.weight", n, n); w(prefix + ".value.bias", n)
        w(prefix + ".out.weight", n, n); w(prefix + ".out.bias", n)

    def mlp(prefix, n):
        w(prefix + ".0.weight", 4 * n, n); w(prefix + ".0.bias", 4 * n)
        w(prefix + ".2.weight", n, 4 * n); w(prefix + ".2.bias", n)

    na, nt, n_mels = dims["n_audio_state"], dims["n_text_state"], dims["n_mels"]
    w("encoder.conv1.weight", na, n_mels, 3, s=1.0 / math.sqrt(3 * n_mels)); w("encoder.conv1.bias", na)
    w("encoder.conv2.weight", na, na, 3, s=1.0 / math.sqrt(3 * na)); w("encoder.conv2.bias", na)
    specs.append(("encoder.positional_embedding", (dims["n_audio_ctx"], na), "pe", 0.0))
    for i in range(dims["n_audio_layer"]):
        p = f"encoder.blocks.{i}"
        ln(p + ".attn_ln", na); attn(p + ".attn", na); ln(p + ".mlp_ln", na); mlp(p + ".mlp", na)
    ln("encoder.ln_post", na)
    w("decoder.token_embedding.weight", dims["n_vocab"], nt, s=logit_std / math.sqrt(nt))
    w("decoder.positional_embedding", dims["n_text_ctx"], nt, s=logit_std / math.sqrt(nt))
    for i in range(dims["n_text_layer"]):
        p = f"decoder.blocks.{i}"
        ln(p + ".attn_ln", nt); attn(p + ".attn", nt)
        ln(p + ".cross_attn_ln", nt); attn(p + ".cross_attn", nt)
        ln(p + ".mlp_ln", nt); mlp(p + ".mlp", nt)
    ln("decoder.ln", nt)
    return specs


def synthetic_state_dict(dims: dict, seed: int = 0, gain: float = 2.0, logit_std: float = 1.5,
                         device=None) -> Dict[str, torch.Tensor]:
    """Tensor number i is filled from its own PCG64 stream seeded with [seed, i], so tensors can be
    generated in parallel threads (1.5e9 values for large-v2) and the result does not depend on
    the number of threads."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    specs = _tensor_specs(dims, gain, logit_std)
    if device is not None and torch.device(device).type == "cuda":
        # benchmark path: same distributions drawn by the GPU's generator (seconds for large-v2);
        # NOT bit-identical to the host streams, so parity tests never use it
        out = {}
        for idx, (name, shape, kind, scale) in enumerate(specs):
            if kind == "pe":
                out[name] = _sinusoids(shape[0], shape[1]).half().to(device)
                continue
            g = torch.Generator(device=device).manual_seed(seed * 1000003 + idx)
            x = torch.empty(shape, dtype=torch.float32, device=device)
            if kind == "n":
                x.normal_(0.0, scale, generator=g)
            else:
                x.uniform_(-scale, scale, generator=g)
                if kind == "g":
                    x += 1.0
            out[name] = x.half()
        return out

    def fill(item):
        idx, (name, shape, kind, scale) = item
        if kind == "pe":
            return name, _sinusoids(shape[0], shape[1]).half()
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        if kind == "n":
            x = rng.standard_normal(shape, dtype=np.float32)
            x *= np.float32(scale)
        else:
            x = rng.random(shape, dtype=np.float32)
            x *= np.float32(2 * scale)
            x += np.float32((1.0 if kind == "g" else 0.0) - scale)
        return name, torch.from_numpy(x.astype(np.float16))

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        return dict(ex.map(fill, enumerate(specs)))


def synthetic_checkpoint(model: str = "large-v2", seed: int = 0, device=None) -> dict:
    dims = dict(DIMS[model]) if isinstance(model, str) else dict(model)
    return {"dims": dims, "model_state_dict": synthetic_state_dict(dims, seed, device=device)}


def synthetic_mel(batch: int, n_frames: int = 3000, n_mels: int = 80, seed: int = 1234) -> torch.Tensor:
    """N(0, 0.5) clipped to [-0.5, 1.5]: the range the reference's log-mel normalisation produces
    (W/whisper_utils.py:143-145).  fp16 [batch, n_mels, n_frames]."""
    rng = np.random.Generator(np.random.PCG64([seed, batch]))
    x = np.clip(rng.standard_normal((batch, n_mels, n_frames), dtype=np.float32) * np.float32(0.5), -0.5, 1.5)
    return torch.from_numpy(x.astype(np.float16))
