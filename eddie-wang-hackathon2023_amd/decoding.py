"""WhisperDecoding: decoder + cross-K/V session wrapper and Whisper's greedy decoding rules.

Same classes, constructor, attributes and method names as the reference's
examples/whisper/decoding.py (W/decoding.py; W = /root/reference/tensorrt_llm_july-release-v1/
examples/whisper): `DecodingOptions`, `DecodingResult`, `MaximumLikelihoodRanker`,
`SuppressBlank`, `SuppressTokens`, `ApplyTimestampRules`, `GreedyDecoder`, and
`WhisperDecoding.{xa2cross_key_value, decode, detect_language, main_loop, post_process,
torch_detect_language, torch_main_loop}` (W/decoding.py:303-878).

What differs underneath:
* any batch size and model size (the reference hard-wires batch 1 / large-v2);
* `main_loop` runs the fast path: KV cache pre-allocated `[B,2,H,n_text_ctx,64]` and appended in
  place (the reference concatenates a fresh cache per layer per step), logit rules +
  arg-max + log-prob + token append fused in one device kernel (csrc/greedy.hip), no host
  synchronisation inside the loop except a completion poll every `poll_every` steps.
  `main_loop_reference` is the literal per-step loop through `decode()` and the host-side filter
  classes, kept for parity tests: both must return identical tokens;
* the cross-attention K/V computed by `detect_language` are re-used by `main_loop` for the same
  audio features (the reference runs that engine twice per utterance).
"""
from __future__ import annotations

import json
import os
import weakref
import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from functools import lru_cache
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import ctypes as C
import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

import native
from build import get_engine_name
from session import Session, TensorInfo, str_dtype_to_trt, trt_dtype_to_torch, logger
from tokenizer import LANGUAGES, TO_LANGUAGE_CODE, Tokenizer

CHUNK_LENGTH = 30


@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"            # "transcribe" (X->X) or "translate" (X->English)
    language: Optional[str] = None      # detected when None
    temperature: float = 0.0
    sample_len: Optional[int] = None    # maximum number of tokens to sample
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    prompt: Optional[Union[str, List[int]]] = None
    prefix: Optional[Union[str, List[int]]] = None
    suppress_tokens: Optional[Union[str, Iterable[int]]] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = True


@dataclass(frozen=True)
class DecodingResult:
    audio_features: Tensor
    language: str
    language_probs: Optional[Dict[str, float]] = None
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = np.nan
    no_speech_prob: float = np.nan
    temperature: float = np.nan
    compression_ratio: float = np.nan


class MaximumLikelihoodRanker:
    """Pick the sample with the highest length-normalised log-probability (W/decoding.py:92-115)."""

    def __init__(self, length_penalty: Optional[float]):
        self.length_penalty = length_penalty

    def rank(self, tokens: List[List[Tensor]], sum_logprobs: List[List[float]]):
        picks = []
        for group, lps in zip(tokens, sum_logprobs):
            scores = []
            for t, lp in zip(group, lps):
                n = len(t)
                penalty = n if self.length_penalty is None else ((5 + n) / 6) ** self.length_penalty
                scores.append(lp / penalty)
            picks.append(int(np.argmax(scores)))
        return picks


# ---- host-side logit filters (the reference-faithful loop; vectorised over the batch) -----------

class LogitFilter:
    def apply(self, logits: Tensor, tokens: Tensor) -> None:
        raise NotImplementedError


class SuppressBlank(LogitFilter):
    def __init__(self, tokenizer: Tokenizer, sample_begin: int):
        self.tokenizer, self.sample_begin = tokenizer, sample_begin

    def apply(self, logits: Tensor, tokens: Tensor):
        if tokens.shape[1] == self.sample_begin:
            logits[:, list(self.tokenizer.blank_tokens()) + [self.tokenizer.eot]] = -np.inf


class SuppressTokens(LogitFilter):
    def __init__(self, suppress_tokens: Sequence[int]):
        self.suppress_tokens = list(suppress_tokens)

    def apply(self, logits: Tensor, tokens: Tensor):
        logits[:, self.suppress_tokens] = -np.inf


class ApplyTimestampRules(LogitFilter):
    """Timestamp pairing / monotonicity / initial-timestamp rules (W/decoding.py:134-199)."""

    def __init__(self, tokenizer: Tokenizer, sample_begin: int, max_initial_timestamp_index: Optional[int]):
        self.tokenizer, self.sample_begin = tokenizer, sample_begin
        self.max_initial_timestamp_index = max_initial_timestamp_index

    def apply(self, logits: Tensor, tokens: Tensor):
        tk = self.tokenizer
        tb, eot = tk.timestamp_begin, tk.eot
        V = logits.shape[1]
        if tk.no_timestamps is not None:
            logits[:, tk.no_timestamps] = -np.inf
        sampled = tokens[:, self.sample_begin:]
        n = sampled.shape[1]
        is_ts = sampled >= tb
        col = torch.arange(V, device=logits.device)[None, :]
        last_ts = is_ts[:, -1] if n >= 1 else torch.zeros(tokens.shape[0], dtype=torch.bool, device=tokens.device)
        pen_ts = is_ts[:, -2] if n >= 2 else torch.ones(tokens.shape[0], dtype=torch.bool, device=tokens.device)
        # timestamps come in pairs (except right before EOT)
        kill = (last_ts & pen_ts)[:, None] & (col >= tb)
        kill |= (last_ts & ~pen_ts)[:, None] & (col < eot)
        if n >= 1:
            # no decreasing timestamps; a closed segment must have non-zero length
            has_ts = is_ts.any(dim=1)
            idx = torch.arange(n, device=tokens.device)[None, :].expand_as(is_ts)
            last_pos = torch.where(is_ts, idx, torch.full_like(idx, -1)).max(dim=1).values.clamp(min=0)
            last_val = sampled.gather(1, last_pos[:, None])[:, 0]
            bound = torch.where(last_ts & ~pen_ts, last_val, last_val + 1)
            kill |= has_ts[:, None] & (col >= tb) & (col < bound[:, None])
        logits.masked_fill_(kill.to(logits.device), -np.inf)
        if tokens.shape[1] == self.sample_begin:
            logits[:, :tb] = -np.inf
            if self.max_initial_timestamp_index is not None:
                logits[:, tb + self.max_initial_timestamp_index + 1:] = -np.inf
        # if the timestamps together outweigh every text token, sample a timestamp
        logprobs = F.log_softmax(logits.float(), dim=-1)
        ts_lp = logprobs[:, tb:].logsumexp(dim=-1)
        text_lp = logprobs[:, :tb].max(dim=-1).values
        logits[:, :tb] = torch.where((ts_lp > text_lp)[:, None], torch.full_like(logits[:, :tb], -np.inf),
                                     logits[:, :tb])


class GreedyDecoder:
    def __init__(self, temperature: float, eot: int):
        self.temperature, self.eot = temperature, eot

    def update(self, tokens: Tensor, logits: Tensor, sum_logprobs: Tensor) -> Tuple[Tensor, bool]:
        if self.temperature == 0:
            next_tokens = logits.argmax(dim=-1)
        else:
            next_tokens = torch.distributions.Categorical(logits=logits / self.temperature).sample()
        logprobs = F.log_softmax(logits.float(), dim=-1)
        current = logprobs[torch.arange(logprobs.shape[0]), next_tokens]
        alive = tokens[:, -1] != self.eot
        sum_logprobs += current * alive
        next_tokens[~alive] = self.eot
        tokens = torch.cat([tokens, next_tokens[:, None]], dim=-1)
        return tokens, bool((tokens[:, -1] == self.eot).all())

    def finalize(self, tokens: Tensor, sum_logprobs: Tensor):
        # make sure each sequence has at least one EOT token at the end
        return F.pad(tokens, (0, 1), value=self.eot), sum_logprobs.tolist()


class WhisperDecoding:
    def __init__(self, engine_dir, only_torch: bool = False, vocab_path: Optional[str] = None,
                 options: Optional[DecodingOptions] = None):
        engine_dir = Path(engine_dir)
        self.decoder_config = None
        self.cross_attn_config = None
        self.get_config(engine_dir)
        self.only_torch = only_torch
        if not only_torch:
            self.decoder_session, self.cross_attn_session = self.get_session(engine_dir)
        self.vocab_path = vocab_path
        multilingual = self.decoder_config.get('vocab_size', 51865) >= 51865
        self.tokenizer = self.get_tokenizer(multilingual, 'en', 'transcribe')
        self.is_multilingual = multilingual

        self.sot_sequence = self.tokenizer.sot_sequence
        self.initial_tokens = tuple(list(self.sot_sequence))
        self.initial_token_length = len(self.initial_tokens)
        self.tokens = torch.tensor([self.initial_tokens]).repeat(self.decoder_config['num_audio'], 1)
        self.sot_index = self.initial_tokens.index(self.tokenizer.sot)
        self.options = options or DecodingOptions()

        self.n_group = self.options.beam_size or self.options.best_of or 1
        self.sample_len: int = self.options.sample_len or self.decoder_config['num_text_ctx'] // 2
        if self.options.prompt or self.options.prefix:
            # the reference carries _get_initial_tokens (W/decoding.py:485-513) but never calls it (its options are the
            # defaults); with caller-supplied options the start sequence is <|startofprev|> prompt <|sot|> .. prefix
            self.initial_tokens = self._get_initial_tokens()
            self.initial_token_length = len(self.initial_tokens)
            self.tokens = torch.tensor([self.initial_tokens]).repeat(self.decoder_config['num_audio'], 1)
            self.sot_index = self.initial_tokens.index(self.tokenizer.sot)
        self.sample_begin: int = len(self.initial_tokens)
        self.use_int8_kv_cache = self.decoder_config['use_int8_kv_cache']
        self.use_int8_cross_kv = bool(self.decoder_config.get('use_int8_cross_kv', False))     # opt-in, beyond the reference

        pe = np.load(engine_dir / 'positional_embedding.npy')
        self.positional_embedding = torch.tensor(pe)
        if not only_torch:
            self.positional_embedding = self.positional_embedding.to('cuda').type(torch.float16).contiguous()

        self.logit_filters = []
        self.max_initial_timestamp_index = None
        if self.options.suppress_blank:
            self.logit_filters.append(SuppressBlank(self.tokenizer, self.sample_begin))
        if self.options.suppress_tokens:
            self.logit_filters.append(SuppressTokens(self._get_suppress_tokens()))
        if not self.options.without_timestamps:
            precision = CHUNK_LENGTH / self.decoder_config['num_audio_ctx']  # usually 0.02 seconds
            if self.options.max_initial_timestamp:
                self.max_initial_timestamp_index = round(self.options.max_initial_timestamp / precision)
            self.logit_filters.append(
                ApplyTimestampRules(self.tokenizer, self.sample_begin, self.max_initial_timestamp_index))
        self.sequence_ranker = MaximumLikelihoodRanker(self.options.length_penalty)
        self.decoder = GreedyDecoder(self.options.temperature, self.tokenizer.eot)

        self.kv_cache = {}
        self.hooks = []
        self._cross_cache = None          # (key, list of cross K/V) from the last xa2cross_key_value
        self._state = {}                  # per-batch-size device buffers of the fast path
        self.poll_every = 8
        self.micro_batches = None         # stream-parallel utterance groups: None = by batch size (_groups), or a fixed count
        self.lang_id_sequential = False   # bench.py: run the language pass group by group (0.2 % of a step) so that its
                                          # HIP-event kernel timings are not inflated by the other group's HBM share
        self.groups_sequential = False    # bench.py's roofline probe: main_loop steps the utterance groups one AFTER the other (each group's
                                          # step waits for the previous group's), so that a launch is timed with the chip to itself
        # experimental schedule (off by default): cross-attention on its own CU set
        # (wm_decoder_step_multi).  Steady state 12.9 ms/step vs 13.9 for the captured graphs at B = 256, but the
        # cross-queue event hand-offs cost 12 us each and only resolve quickly while the host is busy issuing.
        self.cu_partition = False
        self.light_cus = 96               # CUs reserved for the latency-bound kernels of all groups
        self.run_ahead = 3                # decode steps the host may be ahead of the GPU in the partitioned loop
        self._partition = None            # (light streams, heavy stream), created on first use
        self.use_graphs = True            # replay one captured decode step per token (hipGraph)
        self.device_sampling = True       # temperature > 0 / best_of > 1 through the fused device loop (False: the literal host loop)
        # Per-row completion (W/decoding.py:817-819 stops its single utterance at EOT): rows that have emitted EOT drop out
        # of the attention kernels (their cross K/V -- 245.76 MB per token at large-v2 -- and KV cache are no longer read)
        # and a group whose rows have all finished is no longer stepped.  Results are unchanged: a finished row is kept at
        # EOT whatever its logits.  Off for `ignore_eot` loops (benchmarks decode a fixed number of tokens).
        self.skip_finished_rows = True
        self.profile_eager_passes = False  # bench.py's roofline probe: keep the language pass eager (its launches carry the profiling events)
        self.graph_prefill = True         # replay the language pass and the prefill from captured graphs too (they are host-bound eagerly)
        self.force_not_alone = False      # bench.py's probes step the groups one after the other but must run the kernels of the parallel schedule
        self.first_token_event = None     # bench.py: an event recorded (current stream) when the first sampled token of a main_loop call exists
        self._streams = []
        self._no_dedicated_queues = False
        WhisperDecoding._instances.add(self)      # (weak: a give-up of a one-launch step drops the graphs of EVERY instance, _chain_gave_up)

    # ---- configuration / sessions -----------------------------------------------------------------
    def get_config(self, engine_dir):
        for attr, fname in (('decoder_config', 'decoder_config.json'), ('cross_attn_config', 'cross_attn_config.json')):
            with open(engine_dir / fname, 'r') as f:
                config = json.load(f)
            merged = OrderedDict()
            merged.update(config['plugin_config'])
            merged.update(config['builder_config'])
            setattr(self, attr, merged)

    def get_session(self, engine_dir):
        path = engine_dir / get_engine_name('whisper_decoder', self.decoder_config['precision'],
                                            self.decoder_config['tensor_parallel'], 0)
        with open(path, 'rb') as f:
            decoder_session = Session.from_serialized_engine(f.read())
        path = engine_dir / get_engine_name('whsiper_crossattn', self.cross_attn_config['precision'],
                                            self.cross_attn_config['tensor_parallel'], 0)
        with open(path, 'rb') as f:
            cross_attn_session = Session.from_serialized_engine(f.read())
        return decoder_session, cross_attn_session

    def _get_suppress_tokens(self) -> Tuple[int]:
        suppress_tokens = self.options.suppress_tokens
        if isinstance(suppress_tokens, str):
            suppress_tokens = [int(t) for t in suppress_tokens.split(",")]
        if -1 in suppress_tokens:
            suppress_tokens = [t for t in suppress_tokens if t >= 0]
            suppress_tokens.extend(self.tokenizer.non_speech_tokens)
        elif suppress_tokens is None or len(suppress_tokens) == 0:
            suppress_tokens = []
        else:
            assert isinstance(suppress_tokens, list), "suppress_tokens must be a list"
        tk = self.tokenizer
        suppress_tokens.extend([tk.transcribe, tk.translate, tk.sot, tk.sot_prev, tk.sot_lm])
        if tk.no_speech is not None:
            suppress_tokens.append(tk.no_speech)       # no-speech probability is collected separately
        return tuple(sorted(set(suppress_tokens)))

    def get_tokenizer(self, multilingual: bool, language: Optional[str] = None, task: Optional[str] = None) -> Tokenizer:
        path = self.vocab_path
        if not path:      # the vendored vocabulary (assets/ASSETS.md), like the reference's assets/ (W/decoding.py:423-431)
            bundled = Path(__file__).resolve().parent / 'assets' / ('multilingual.tiktoken' if multilingual else 'gpt2.tiktoken')
            path = str(bundled) if bundled.exists() else None
        if path:
            return Tokenizer.from_vocab(path, multilingual, language, task)
        return Tokenizer.ids_only(multilingual, language, task)

    def _get_initial_tokens(self) -> Tuple[int]:
        tokens = list(self.sot_sequence)
        if prefix := self.options.prefix:
            prefix_tokens = self.tokenizer.encode(" " + prefix.strip()) if isinstance(prefix, str) else prefix
            if self.sample_len is not None:
                max_prefix_len = self.decoder_config['num_text_ctx'] // 2 - self.sample_len
                prefix_tokens = prefix_tokens[-max_prefix_len:]
            tokens = tokens + prefix_tokens
        if prompt := self.options.prompt:
            prompt_tokens = self.tokenizer.encode(" " + prompt.strip()) if isinstance(prompt, str) else prompt
            tokens = [self.tokenizer.sot_prev] + prompt_tokens[-(self.decoder_config['num_text_ctx'] // 2 - 1):] + tokens
        return tuple(tokens)

    # ---- engine calls with the reference's by-name protocol ------------------------------------------
    @staticmethod
    def _features_key(xa):
        """Identity of an audio-features tensor's CONTENT, or None when it cannot be known: the generation number
        the encoder stamped on it (encoding.stamp_generation) plus pointer, shape and torch's version counter.
        Pointer and version alone are not enough -- the engine writes through raw pointers, so a second batch
        encoded into the same buffer looks unchanged to torch.  A tensor without a stamp (a slice, a clone, a
        tensor from elsewhere) is never assumed to be one seen before."""
        gen = getattr(xa, 'wm_generation', None)
        return None if gen is None else (gen, xa.data_ptr(), tuple(xa.shape), xa._version)

    def xa2cross_key_value(self, xa):
        key = self._features_key(xa)
        if key is not None and self._cross_cache is not None and self._cross_cache[0] == key:
            return self._cross_cache[1]
        inputs = OrderedDict()
        xa16 = xa.type(torch.float16).contiguous()
        inputs.update({'xa': xa16})
        output_info = self.cross_attn_session.infer_shapes([TensorInfo('xa', str_dtype_to_trt("float16"), xa16.shape)])
        logger.debug(f'output info {output_info}')
        outputs = {t.name: torch.empty(tuple(t.shape), dtype=trt_dtype_to_torch(t.dtype), device=xa.device)
                   for t in output_info}
        stream = torch.cuda.current_stream()
        ok = self.cross_attn_session.run(inputs=inputs, outputs=outputs, stream=stream.cuda_stream)
        assert ok, 'Engine execution failed'
        stream.synchronize()
        cross = [outputs['cross_present_key_value_' + str(i)] for i in range(self.cross_attn_config['num_layers'])]
        self._cross_cache = (key, cross, xa)      # keep xa alive so the pointer key stays valid
        return cross

    def decode(self, x, cross_past_key_value, past_key_value=None):
        """One decoder call with the reference's I/O protocol (W/decoding.py:543-659): returns
        (logits fp16 [B, L, n_vocab], list of present_key_value [B,2,H,T+L,64])."""
        dev = x.device
        n_layer = self.decoder_config['num_layers']
        kv_dtype = 'int8' if self.use_int8_kv_cache else 'float16'
        inputs = OrderedDict()
        infos = []

        def add(name, tensor, dtype, shape=None):
            inputs[name] = tensor
            infos.append(TensorInfo(name, str_dtype_to_trt(dtype), tuple(shape if shape is not None else tensor.shape)))

        x = x.type(torch.int32).contiguous()
        input_len = x.shape[-1]
        add('x', x, 'int32')
        lens = torch.tensor((input_len,), dtype=torch.int32, device=dev)
        add('input_lengths', lens, 'int32')
        add('max_input_length', lens, 'int32')
        if not self.decoder_config['gpt_attention_plugin']:
            n_ctx = self.decoder_config['num_text_ctx']
            mask = torch.full((n_ctx, n_ctx), -50000.0, dtype=torch.float32).triu_(1)[:input_len, :input_len].contiguous().to(dev)
            add('mask', mask, 'float32')          # causality is applied inside the kernel; kept for I/O parity
        else:
            add('masked_tokens', torch.zeros((1, input_len), dtype=torch.int32, device=dev), 'int32')
            add('cache_indirection', torch.zeros((x.shape[0], 1, input_len), dtype=torch.int32, device=dev), 'int32')
            add('past_key_value_length', torch.tensor([0, 1], dtype=torch.int32, device=dev), 'int32')
            add('sequence_length', lens, 'int32')
        offset = past_key_value[0].shape[3] if past_key_value else 0
        pos = self.positional_embedding[offset: offset + input_len].type(torch.float16).contiguous()
        add('positional_embedding', pos, 'float16')
        n_head = self.decoder_config['num_heads']
        for i in range(n_layer):
            if past_key_value is None:
                dummy = torch.ones((1,), dtype=trt_dtype_to_torch(kv_dtype), device=dev)
                add('past_key_value_' + str(i), dummy, kv_dtype, (x.shape[0], 2, n_head, 0, 64))
            else:
                add('past_key_value_' + str(i), past_key_value[i].contiguous(), kv_dtype)
        for i in range(n_layer):
            add('cross_past_key_value_' + str(i), cross_past_key_value[i].contiguous(),
                'int8' if self.use_int8_cross_kv else 'float16')

        output_info = self.decoder_session.infer_shapes(infos)
        assert output_info is not None, 'infer_shapes failed'
        logger.debug(f'output info {output_info}')
        outputs = {t.name: torch.empty(tuple(t.shape), dtype=trt_dtype_to_torch(t.dtype), device=dev) for t in output_info}
        stream = torch.cuda.current_stream()
        ok = self.decoder_session.run(inputs=inputs, outputs=outputs, stream=stream.cuda_stream)
        assert ok, 'Engine execution failed'
        stream.synchronize()
        # a first call of one token for up to eight utterances (the language pass) may have run as ONE launch (gemv_chain.hip): this path
        # has synchronised, so a look at the give-up word is free -- and nobody else would look (the caller decoded on from garbage)
        if x.is_cuda and input_len == 1 and x.shape[0] <= 8 and self._chain_gave_up("decode()"):
            ok = self.decoder_session.run(inputs=inputs, outputs=outputs, stream=stream.cuda_stream)     # launch per kernel now
            assert ok, 'Engine execution failed'
            stream.synchronize()
        return outputs['output'], [outputs['present_key_value_' + str(i)] for i in range(n_layer)]

    # ---- language detection ---------------------------------------------------------------------------
    def _language_from_logits(self, logits, n_audio, single):
        """Mask everything but the language tokens, arg-max and softmax (W/decoding.py:724-735).  Same arithmetic as the
        reference (softmax over the full, masked row); only the language columns travel to the host, as one block --
        one `.item()` per (utterance, language) took 100 ms per batch of 576."""
        tk = self.tokenizer
        lang_tokens, codes = list(tk.all_language_tokens), list(tk.all_language_codes)
        key = (str(logits.device), logits.shape[-1])
        cache = getattr(self, "_lang_mask_cache", None)
        if cache is None or cache[0] != key:
            mask = torch.ones(logits.shape[-1], dtype=torch.bool)
            mask[lang_tokens] = False
            cache = (key, mask.to(logits.device), torch.tensor(lang_tokens, dtype=torch.long, device=logits.device))
            self._lang_mask_cache = cache
        _, mask_dev, idx = cache
        logits.masked_fill_(mask_dev, -np.inf)
        language_tokens = logits.argmax(dim=-1)
        probs = logits.softmax(dim=-1).index_select(-1, idx).cpu().tolist()
        language_probs = [dict(zip(codes, row)) for row in probs]
        if single:
            language_tokens, language_probs = language_tokens[0], language_probs[0]
            languages = [max(language_probs, key=language_probs.get)]
        else:
            languages = [max(p, key=p.get) for p in language_probs]
        return language_tokens, language_probs, languages

    def detect_language_reference(self, audio_features):
        """The reference's language-ID pass literally (W/decoding.py:703-741): cross K/V engine, one
        `decode()` of the single <|sot|> token through the by-name protocol."""
        if not self.is_multilingual:
            return ['en'] * audio_features.shape[0], None      # English-only vocabulary: no language tokens
        languages = [self.options.language] * audio_features.shape[0]
        language_probs = None
        if self.options.language is None or self.options.task == "lang_id":
            single = audio_features.ndim == 2
            if single:
                audio_features = audio_features.unsqueeze(0)
            n_audio = audio_features.shape[0]
            x = torch.tensor([[self.tokenizer.sot]] * n_audio).to(audio_features.device)    # [n_audio, 1]
            cross = self.xa2cross_key_value(audio_features)
            logits, _ = self.decode(x, cross)
            language_tokens, language_probs, languages = self._language_from_logits(logits[:, 0].float(), n_audio, single)
            if self.options.language is None:
                self.tokens = torch.tensor([self.initial_tokens]).repeat(n_audio, 1)
                self.tokens[:, self.sot_index + 1] = language_tokens.cpu()        # write language tokens
        return languages, language_probs

    def detect_language(self, audio_features, _retry: bool = False):
        """Language-ID pass, fast path: same arithmetic as detect_language_reference, but the cross
        K/V land in the persistent buffers main_loop re-uses (the reference computes them twice,
        SURVEY F6) and the one-token decoder call runs per utterance group on its stream."""
        if not self.is_multilingual:
            return ['en'] * audio_features.shape[0], None      # English-only vocabulary: no language tokens
        languages = [self.options.language] * audio_features.shape[0]
        language_probs = None
        if self.options.language is None or self.options.task == "lang_id":
            single = audio_features.ndim == 2
            if single:
                gen = getattr(audio_features, 'wm_generation', None)
                audio_features = audio_features.unsqueeze(0)
                if gen is not None:          # the view is the same encoder output: it keeps the stamp (_features_key; else the cross K/V
                    audio_features.wm_generation = gen      # of one clip would be projected twice, once per view)
            n_audio, dev = audio_features.shape[0], audio_features.device
            cfg = self.decoder_config
            if self.n_group > 1 and self.device_sampling and audio_features.is_cuda:
                # best_of / beam candidates: main_loop decodes n_audio x n_group rows.  The state keeps ONE buffer set at a time, so a
                # language pass over n_audio rows would free and re-allocate the caches, the cross K/V buffers and the graphs on every
                # clip.  The pass runs over the candidates' rows instead -- the SAME repeated tensor main_loop will use
                # (_candidate_rows): one buffer set, the cross K/V computed once for both calls -- and reads every n_group-th row
                languages_r, probs_r = self._detect_language_rows(self._candidate_rows(audio_features), False, _retry)
                languages = languages_r[:: self.n_group]
                language_probs = probs_r[:: self.n_group] if probs_r is not None else None
                if self.options.language is None:
                    self.tokens = self.tokens[:: self.n_group].contiguous()
                if single:
                    language_probs = language_probs[0] if language_probs is not None else None
                return languages, language_probs
            return self._detect_language_rows(audio_features, single, _retry)
        return languages, language_probs

    def _candidate_rows(self, audio_features):
        """The audio features with every clip repeated n_group times (best_of / beam candidates share their utterance's audio: the
        reference repeats the features too, W/decoding.py:538-539) -- ONE tensor per encoder run, so that the language pass and the
        decode loop key the same cross K/V (`_features_key`) and the same buffer set."""
        from encoding import stamp_generation
        key = self._features_key(audio_features)
        c = getattr(self, "_rep_cache", None)
        if key is not None and c is not None and c[0] == key:
            return c[1]
        rep = stamp_generation(audio_features.repeat_interleave(self.n_group, dim=0))
        self._rep_cache = (key, rep, audio_features)       # (the original kept alive: its pointer is part of the key)
        return rep

    def _detect_language_rows(self, audio_features, single, _retry):
        """The pass itself over the rows of `audio_features` [n, n_audio_ctx, C] (detect_language)."""
        languages, language_probs = None, None
        n_audio, dev = audio_features.shape[0], audio_features.device
        cfg = self.decoder_config
        st = self._fast_state(n_audio, dev)
        n_micro_, bounds_ = self._groups(n_audio)   # groups of up to eight rows stepped one at a time may run as ONE launch per step
        one_row = (n_micro_ == 1 or self.lang_id_sequential) and not self.force_not_alone and any(hi - lo <= 8 for lo, hi in bounds_)
        if one_row and native.chain_status()["error_pending"]:                  # (a peek at a host word: no synchronisation)
            self._chain_gave_up("found before the language pass", foreign=True)  # somebody else's give-up: said loudly, acknowledged, not ours to repeat
        cross = self._cross_persistent(audio_features, st)
        if 'lang_logits' not in st:
            st['lang_logits'] = torch.empty((n_audio, 1, cfg['vocab_size']), dtype=torch.float16, device=dev)
            st['sot'] = torch.full((n_audio, 1), self.tokenizer.sot, dtype=torch.int32, device=dev)
        n_micro, bounds = self._groups(n_audio)
        main = torch.cuda.current_stream()
        streams = self._group_streams(n_micro, dev)
        cap = cfg['num_text_ctx']
        use_graph = self.use_graphs and self.graph_prefill and self.decoder_session.qkv_amax is None and not self.profile_eager_passes
        for g, (lo, hi) in enumerate(bounds):
            streams[g].wait_stream(main)
            if self.lang_id_sequential and g > 0:
                streams[g].wait_stream(streams[g - 1])     # one group at a time: kernels are timed un-shared (bench.py)
            na = (n_micro > 1 and not self.lang_id_sequential) or self.force_not_alone

            def issue_lang(g=g, lo=lo, hi=hi, na=na):
                self.decoder_session.decoder_step(st['sot'][lo:hi], self.positional_embedding[0:1],
                                                  [t[lo:hi] for t in cross], None, cap, [t[lo:hi] for t in st['kv']], cap,
                                                  st['lang_logits'][lo:hi], 0, streams[g].cuda_stream, slot=g, not_alone=na)
            # a launch per kernel (more than eight rows, or groups side by side): ~ 290 eager launches per group -> a replayed graph from the
            # second batch on.  (Up to eight rows alone the pass is the one-launch step: four launches, nothing to replay.)
            lkey = (n_micro, g, 'lang', na)
            if use_graph and (na or hi - lo > 8) and lkey in st['graphs']:
                with torch.cuda.stream(streams[g]):
                    st['graphs'][lkey].replay()
            else:
                issue_lang()
                if use_graph and (na or hi - lo > 8):
                    self._capture(st, lkey, streams[g], issue_lang)
            main.wait_stream(streams[g])
        language_tokens, language_probs, languages = self._language_from_logits(
            st['lang_logits'][:, 0].float(), n_audio, single)                # (brings the logits to the host: the pass has finished)
        if one_row and not _retry and self._chain_gave_up("language pass"):
            return self._detect_language_rows(audio_features, single, True)
        if self.options.language is None:
            self.tokens = torch.tensor([self.initial_tokens]).repeat(n_audio, 1)
            self.tokens[:, self.sot_index + 1] = language_tokens.cpu()        # write language tokens
        return languages, language_probs

    def torch_detect_language(self, model, audio_features):
        """PyTorch path (W/decoding.py:661-701): `model.logits(tokens, audio_features)`."""
        if not self.is_multilingual:
            return ['en'] * audio_features.shape[0], None
        with torch.no_grad():
            languages = [self.options.language] * audio_features.shape[0]
            language_probs = None
            if self.options.language is None or self.options.task == "lang_id":
                single = audio_features.ndim == 2
                if single:
                    audio_features = audio_features.unsqueeze(0)
                n_audio = audio_features.shape[0]
                x = torch.tensor([[self.tokenizer.sot]] * n_audio).to(audio_features.device)
                logits = model.logits(x, audio_features)[:, 0]
                language_tokens, language_probs, languages = self._language_from_logits(logits, n_audio, single)
                if self.options.language is None:
                    self.tokens = torch.tensor([self.initial_tokens]).repeat(n_audio, 1)
                    self.tokens[:, self.sot_index + 1] = language_tokens.cpu()
        for hook in self.hooks:
            hook.remove()
        self.kv_cache, self.hooks = {}, []
        return languages, language_probs

    # ---- decoding loops ---------------------------------------------------------------------------------
    def _initial_token_rows(self, n_audio, device):
        tokens = self.tokens
        if tokens.shape[0] != n_audio:
            tokens = torch.tensor([self.initial_tokens]).repeat(n_audio, 1)
        return tokens.repeat_interleave(self.n_group, dim=0).to(device)

    def _host_step(self, i, logits, tokens, sum_logprobs, no_speech_probs):
        if i == 0 and self.tokenizer.no_speech is not None:       # save no_speech_probs
            probs_at_sot = logits[:, self.sot_index].float().softmax(dim=-1)
            no_speech_probs[:] = probs_at_sot[:, self.tokenizer.no_speech].tolist()
        logits = logits[:, -1]
        for logit_filter in self.logit_filters:
            logit_filter.apply(logits, tokens)
        return self.decoder.update(tokens, logits, sum_logprobs)

    def main_loop_reference(self, audio_features):
        """The reference's loop verbatim in structure (W/decoding.py:785-821): one `decode()` per token,
        host-side filters, concat KV."""
        tokens = self._initial_token_rows(audio_features.shape[0], audio_features.device)
        n_batch = tokens.shape[0]
        sum_logprobs: Tensor = torch.zeros(n_batch, device=audio_features.device)
        no_speech_probs = [np.nan] * n_batch
        past_key_value = None
        if self.n_group > 1:        # best_of / beam candidates share their utterance's audio (upstream Whisper repeats it too)
            audio_features = audio_features.repeat_interleave(self.n_group, dim=0)
        cross = self.xa2cross_key_value(audio_features)
        for i in range(self.sample_len):
            feed = tokens if tokens.shape[-1] <= self.initial_token_length else tokens[:, -1:]
            logits, past_key_value = self.decode(feed, cross, past_key_value)
            tokens, completed = self._host_step(i, logits, tokens, sum_logprobs, no_speech_probs)
            if completed or tokens.shape[-1] > self.decoder_config['num_text_ctx']:
                break
        return tokens, sum_logprobs, no_speech_probs

    def torch_main_loop(self, model, audio_features):
        """PyTorch path (W/decoding.py:743-783): `model.decoder(tokens, xa, kv_cache=)` with hooks."""
        with torch.no_grad():
            tokens = self._initial_token_rows(audio_features.shape[0], audio_features.device)
            n_batch = tokens.shape[0]
            sum_logprobs: Tensor = torch.zeros(n_batch, device=audio_features.device)
            no_speech_probs = [np.nan] * n_batch
            for i in range(self.sample_len):
                if not self.kv_cache:
                    self.kv_cache, self.hooks = model.install_kv_cache_hooks()
                feed = tokens if tokens.shape[-1] <= self.initial_token_length else tokens[:, -1:]
                logits = model.decoder(feed, audio_features, kv_cache=self.kv_cache)
                tokens, completed = self._host_step(i, logits, tokens, sum_logprobs, no_speech_probs)
                if completed or tokens.shape[-1] > self.decoder_config['num_text_ctx']:
                    break
        for hook in self.hooks:
            hook.remove()
        self.kv_cache, self.hooks = {}, []
        return tokens, sum_logprobs, no_speech_probs

    # ---- fast path ----------------------------------------------------------------------------------------
    def _fast_state(self, n_batch, device):
        st = self._state.get(n_batch)
        if st is not None:
            return st
        # One buffer set at a time: a set holds the KV cache, the persistent cross K/V (245.76 MB per utterance at
        # large-v2: 141 GB at B = 576) and the captured graphs that point into them; a ragged last batch must not
        # allocate a second set next to the first.  Graphs go first (they reference the buffers).
        for old in self._state.values():
            old['graphs'].clear()
        self._state.clear()
        if device.type == 'cuda':
            torch.cuda.empty_cache()
        cfg = self.decoder_config
        n_layer, n_head, cap, V = cfg['num_layers'], cfg['num_heads'], cfg['num_text_ctx'], cfg['vocab_size']
        kv_dtype = torch.int8 if self.use_int8_kv_cache else torch.float16
        if device.type == 'cuda':
            # the batch ceiling is a property of the free memory, said up front instead of an allocator error half way through
            per_row = self.state_bytes_per_utterance()
            free, _ = torch.cuda.mem_get_info(device)
            if n_batch * per_row > free:
                raise RuntimeError(
                    f"WhisperDecoding: a batch of {n_batch} utterances needs {n_batch * per_row / 2**30:.1f} GiB of decoder state "
                    f"({per_row / 2**20:.1f} MiB per utterance: self-attention cache + cross-attention K/V of {n_layer} layers), "
                    f"{free / 2**30:.1f} GiB are free on {device}: at most {int(free // per_row)} utterances per batch fit "
                    f"(another WhisperDecoding's buffer set in this process counts against it)")
        tk = self.tokenizer
        suppress = []
        for f in self.logit_filters:
            if isinstance(f, SuppressTokens):
                suppress += list(f.suppress_tokens)
        if not self.options.without_timestamps and tk.no_timestamps is not None:
            suppress.append(tk.no_timestamps)
        suppress = sorted(set(suppress))
        blank = list(tk.blank_tokens()) + [tk.eot] if self.options.suppress_blank else []
        st = dict(
            kv=[torch.zeros((n_batch, 2, n_head, cap, 64), dtype=kv_dtype, device=device) for _ in range(n_layer)],
            cross=[torch.empty((n_batch, 2, n_head, cfg['num_audio_ctx'], 64),
                               dtype=torch.int8 if self.use_int8_cross_kv else torch.float16, device=device)
                   for _ in range(n_layer)],
            cross_key=None, cross_xa=None, graphs={}, counters={},
            tokens=torch.zeros((n_batch, cap + 1), dtype=torch.int32, device=device),
            logits=torch.empty((n_batch, self.initial_token_length, V), dtype=torch.float16, device=device),
            sum_logprobs=torch.zeros(n_batch, dtype=torch.float32, device=device),
            n_done=torch.zeros(1, dtype=torch.int32, device=device),
            done=torch.zeros(n_batch, dtype=torch.int32, device=device),          # per row: 1 once it has emitted EOT
            seed=torch.zeros(2, dtype=torch.int32, device=device),                # the sampling generator's seed (temperature > 0)
            live={},
            # per-row sample_len (stable address: the captured graphs read it); 2^30 = no limit
            row_limit=torch.full((n_batch,), 1 << 30, dtype=torch.int32, device=device),
            suppress=torch.tensor(suppress or [0], dtype=torch.int32, device=device), n_suppress=len(suppress),
            blank=torch.tensor(blank or [0], dtype=torch.int32, device=device), n_blank=len(blank),
        )
        self._state[n_batch] = st
        return st

    def state_bytes_per_utterance(self) -> int:
        """Device bytes of decoder state one utterance of a batch holds (what `_fast_state` allocates per row): the
        self-attention cache [2, H, n_text_ctx, 64] and the cross-attention K/V [2, H, n_audio_ctx, 64] of every layer
        (245.76 MB at large-v2 with fp16 cross K/V), tokens and the prefill logits."""
        cfg = self.decoder_config
        n_layer, n_head, cap, V = cfg['num_layers'], cfg['num_heads'], cfg['num_text_ctx'], cfg['vocab_size']
        kv = 2 * n_head * cap * 64 * (1 if self.use_int8_kv_cache else 2)
        cross = 2 * n_head * cfg['num_audio_ctx'] * 64 * (1 if self.use_int8_cross_kv else 2)
        return n_layer * (kv + cross) + (cap + 1) * 4 + self.initial_token_length * V * 2 + 64

    def _group_streams(self, n, dev):
        """Side streams of the utterance groups (never the legacy default stream: graphs are captured on them).

        Each group gets a HARDWARE QUEUE OF ITS OWN: a stream created with a full CU mask
        (wm_stream_create_cu_mask) is never multiplexed, whereas ordinary streams share ROCm's pool of
        GPU_MAX_HW_QUEUES (4) queues and which of them end up sharing depends on how many streams the process
        created before.  Measured (B = 384, three groups): 18.1 ms per decode step with a queue per group, 23.5 ms when
        two groups happen to share one -- e.g. in any process that used one more stream earlier (RCCL, a calibration
        pass).  WM_DEDICATED_QUEUES=0 falls back to torch's stream pool."""
        if os.environ.get("WM_DEDICATED_QUEUES", "1") != "0" and not self._no_dedicated_queues:
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            try:
                with torch.cuda.device(dev):
                    return [native.create_masked_stream([True] * n_cu, k) for k in range(n)]
            except native.WmError as e:       # a scheduling nicety, not part of the compute path: fall back to pooled streams
                logger.warning("dedicated hardware queues unavailable (%s): utterance groups use pooled streams", e)
                self._no_dedicated_queues = True
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(device=dev))
        return self._streams[:n]

    def _partition_streams(self, n, dev):
        """CU-partitioned streams for wm_decoder_step_multi: `n` light streams that may only use the first
        `light_cus` CUs and one heavy stream that owns the rest.  Measured on MI355X (scripts/cumask_probe.py):
        192 CUs still pull 6.4 TB/s through the cross-attention kernel, 64 CUs run the weight-streaming chain
        at its full-chip speed, and only with disjoint CU sets do the two actually run at the same time."""
        if self._partition is None or len(self._partition[0]) < n:
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            light_cus = max(8, min(self.light_cus, n_cu // 2))
            light_mask = [i < light_cus for i in range(n_cu)]
            heavy_mask = [not b for b in light_mask]
            with torch.cuda.device(dev):
                light = [native.create_masked_stream(light_mask, k) for k in range(n)]
                heavy = native.create_masked_stream(heavy_mask, n)
            self._partition = (light, heavy)
        return self._partition[0][:n], self._partition[1]

    def _groups(self, n_batch):
        # three groups from 128 utterances up (two chains of short kernels hide under the third group's K/V stream;
        # four concurrent chains are slower again: 18.1 / 28.0 ms per step at B = 384), two from 13 (round 6: 14 utterances 2.29 against
        # 2.39 ms per token step, 12: 2.15 either way, 10: 2.05 / 2.08; rounds 1-5: from 16), else one
        if self.micro_batches is None:
            n_micro = 3 if n_batch >= 128 else 2 if n_batch >= 13 else 1
        else:
            n_micro = self.micro_batches if n_batch >= 4 * self.micro_batches else 1
        return n_micro, [(g * n_batch // n_micro, (g + 1) * n_batch // n_micro) for g in range(n_micro)]

    def balanced_order(self, n_batch: int) -> List[int]:
        """For a batch whose rows are sorted by expected decode length: the permutation that deals them over the
        utterance groups (contiguous slices of the batch, `_groups`) like cards, so that every group holds short and long
        rows alike.  All groups then shrink together as rows finish -- their K/V streams and chains keep overlapping --
        instead of the group of the longest clips running on alone.  `batch[i] = sorted_batch[order[i]]`."""
        n_micro, bounds = self._groups(n_batch)
        room = [hi - lo for lo, hi in bounds]
        members = [[] for _ in bounds]
        g = 0
        for i in range(n_batch):
            while room[g] == 0:
                g = (g + 1) % n_micro
            members[g].append(i)
            room[g] -= 1
            g = (g + 1) % n_micro
        return [i for m in members for i in m]

    def _cross_persistent(self, xa, st):
        """Cross K/V of `xa` in the state's persistent buffers (stable addresses: the captured decode
        graphs point at them).  Computed once per encoder run (`_features_key`: detect_language and main_loop
        on the same encoder output share them; anything else is recomputed)."""
        key = self._features_key(xa)
        if key is None or st['cross_key'] != key:
            xa16 = xa.type(torch.float16).contiguous()
            self.cross_attn_session.cross_kv(xa16, st['cross'], torch.cuda.current_stream().cuda_stream)
            st['cross_key'], st['cross_xa'] = key, xa
        return st['cross']

    def _capture(self, st, key, stream, issue):
        """Capture what `issue()` enqueues on `stream` (it has just been issued eagerly once, so every buffer and every workspace state it
        touches exists) into st['graphs'][key].  Thread-local error mode + CAPTURE_LOCK: see the token step's capture in main_loop."""
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with native.CAPTURE_LOCK, torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            issue()
        st['graphs'][key] = graph

    def _chain_gave_up(self, where: str, foreign: bool = False) -> bool:
        """Did a one-launch decode step (csrc/gemv_chain.hip: groups of up to eight rows) give up waiting for its workgroups since the last look?
        The launch needs its workgroups resident together; another tenant holding CUs or LDS while it is dispatched makes its bounded
        waits expire, and everything decoded from that step is invalid.  The library then stops using the one-launch forms on this
        device (wm_decode_chain_error: acknowledged, the device takes a launch per kernel from now on); graphs captured with chain
        launches are dropped -- those of EVERY WhisperDecoding of the process (another instance would go on replaying chain launches on
        a device the library has taken off the form).  Returns True when the caller has to decode again.
        `foreign`: the word was found set BEFORE this call issued anything -- an earlier call on this device (another instance, a
        Session.run, a C caller) produced an invalid step and never looked.  That is said at error level EVERY time: acknowledging clears
        the word, and the earlier caller's results stay wrong without anybody else being told."""
        err = C.c_int(0)
        native.check(native.load_library().wm_decode_chain_error(C.byref(err)), "wm_decode_chain_error")
        if not err.value:
            return False
        for inst in list(WhisperDecoding._instances):
            for st in inst._state.values():
                st['graphs'].clear()
        if foreign:
            logger.error("whisper_mi355: a one-launch decode step issued BEFORE this call (%s) gave up waiting for its workgroups and nobody "
                         "looked: whatever that earlier call returned on this device is INVALID (decode it again).  Acknowledged here; the "
                         "device takes a launch per kernel from now on (wm_set_decode_chain re-arms the one-launch step)", where)
        elif not WhisperDecoding._chain_warned:
            WhisperDecoding._chain_warned = True
            logger.warning("whisper_mi355: a one-launch decode step gave up waiting for its workgroups (%s; the GPU was not free to hold them "
                           "together): decoding again with a launch per kernel, which this device uses from now on "
                           "(wm_set_decode_chain re-arms the one-launch step)", where)
        return True

    _chain_warned = False
    _instances = weakref.WeakSet()

    def _live_list(self, st, n_micro, slot, lo, hi):
        """The group's list of rows still decoding (int32 [1 + n]: count, indices relative to the group), reset to
        "all of them"; wm_step_finish rebuilds it after every step from the `done` flags."""
        key = (n_micro, slot)
        if key not in st['live']:
            st['live'][key] = torch.empty(1 + hi - lo, dtype=torch.int32, device=st['done'].device)
        live = st['live'][key]
        live[0] = hi - lo
        live[1:] = torch.arange(hi - lo, dtype=torch.int32, device=live.device)
        return live

    def _greedy(self, st, lo, hi, logits_ptr, row_stride, cur_len, stream, n_past_dev=None):
        """Fused logit rules + arg-max + append for utterances [lo, hi) of the batch state `st`."""
        tk = self.tokenizer
        io = native.WmGreedyIO()
        io.logits, io.row_stride = logits_ptr, row_stride
        io.batch, io.n_vocab = hi - lo, self.decoder_config['vocab_size']
        io.tokens, io.tokens_ld, io.cur_len = st['tokens'][lo:hi].data_ptr(), st['tokens'].shape[1], cur_len
        io.sum_logprobs = st['sum_logprobs'][lo:hi].data_ptr()
        io.suppress, io.n_suppress = st['suppress'].data_ptr(), st['n_suppress']
        io.blank, io.n_blank = st['blank'].data_ptr(), st['n_blank']
        io.sample_begin, io.eot, io.timestamp_begin = self.sample_begin, tk.eot, tk.timestamp_begin
        io.max_initial_timestamp_index = -1 if self.max_initial_timestamp_index is None else self.max_initial_timestamp_index
        # 1: SuppressBlank + SuppressTokens + ApplyTimestampRules; 2: without_timestamps -- the reference then builds no timestamp
        # filter (W/decoding.py:337-346) and samples from the whole suppressed vocabulary
        io.apply_rules = 2 if self.options.without_timestamps else 1
        io.n_done = st['n_done'].data_ptr()
        io.n_past_dev = n_past_dev.data_ptr() if n_past_dev is not None else None
        io.done = st['done'][lo:hi].data_ptr()
        io.row_limit = st['row_limit'][lo:hi].data_ptr()
        # temperature > 0: the draw happens in the same kernel (Gumbel-max, counter-based generator keyed on the global row: lo + b);
        # the seed lives in device memory so that a replayed graph draws afresh in every main_loop call
        io.temperature, io.row0, io.seed_dev = float(self.options.temperature), lo, st['seed'].data_ptr()
        native.check(native.load_library().wm_greedy_step(C.byref(io), stream), "wm_greedy_step")

    def main_loop(self, audio_features, ignore_eot: bool = False, row_limit=None, _retry: bool = False):
        """Greedy decoding, fast path.  Same return values as the reference's main_loop
        (tokens int64 [n, <=n_text_ctx+1], sum_logprobs fp32 [n], no_speech_probs list).
        `ignore_eot` (benchmarks with random weights) decodes `sample_len` tokens regardless.
        `row_limit` (optional, int [n]): a per-utterance `sample_len` -- row b ends with EOT after row_limit[b] sampled
        tokens (e.g. a bound from the clip's duration; bench.py's length distributions).

        Utterances are independent, so the batch is cut into `micro_batches` groups that advance in
        lock-step on separate HIP streams: while one group streams its cross-attention K/V (the
        HBM-bound part of a step) the other group runs its latency-bound weight-streaming chain.
        Rows that have emitted EOT drop out of the attention kernels and a group whose rows are all
        finished is no longer stepped (`skip_finished_rows`): the loop's cost follows the live rows."""
        features_in = audio_features
        if self.options.temperature != 0 or self.n_group != 1:
            # Sampling options (W/decoding.py:274-300 temperature, :92-115 best_of + ranker).  Round 4: the device loop takes them --
            # the draw is a Gumbel-max inside the greedy kernel, candidates are rows like any others.  `device_sampling = False`
            # (or tensors that are not on the GPU) keeps the literal host loop: torch's generator, the draws the goldens hold.
            if not self.device_sampling or not audio_features.is_cuda:
                return self.main_loop_reference(audio_features)
            if self.n_group > 1:      # candidates share their utterance's audio: the reference repeats the features too (:538-539 of this file)
                audio_features = self._candidate_rows(audio_features)
        dev = audio_features.device
        tokens0 = self._initial_token_rows(audio_features.shape[0] // self.n_group, dev)
        n_batch, L0 = tokens0.shape
        assert n_batch == audio_features.shape[0]
        cfg = self.decoder_config
        V, cap = cfg['vocab_size'], cfg['num_text_ctx']
        st = self._fast_state(n_batch, dev)
        n_micro_, bounds_ = self._groups(n_batch)       # groups of up to eight rows stepped one at a time may run as ONE launch per token step
        one_row = (n_micro_ == 1 or self.groups_sequential) and not self.force_not_alone and any(hi - lo <= 8 for lo, hi in bounds_)
        if one_row and not _retry and native.chain_status()["error_pending"]:   # (a peek at a host word: no synchronisation)
            self._chain_gave_up("found before the decode loop", foreign=True)   # somebody else's give-up: said loudly, acknowledged, not ours to repeat
        if self.options.temperature != 0:     # a fresh seed per call from torch's generator: torch.manual_seed makes a run repeatable
            if not _retry:
                st['seed'].copy_(torch.randint(0, 2 ** 31 - 1, (2,), dtype=torch.int32))
        cross = self._cross_persistent(audio_features, st)
        st['tokens'].zero_()
        st['tokens'][:, :L0] = tokens0.to(torch.int32)
        st['sum_logprobs'].zero_()
        st['n_done'].zero_()
        st['done'].zero_()
        if row_limit is not None:
            row_limit = torch.as_tensor(row_limit).to(device=dev, dtype=torch.int32)
            assert row_limit.shape == (n_batch,)
            st['row_limit'].copy_(row_limit)
        else:
            st['row_limit'].fill_(1 << 30)
        n_micro, bounds = self._groups(n_batch)
        if self.cu_partition and n_micro > 1 and self.decoder_session.qkv_amax is None and row_limit is None:
            return self._main_loop_partitioned(audio_features, st, cross, L0, n_micro, bounds, ignore_eot)
        use_live = bool(self.skip_finished_rows) and not ignore_eot and max(hi - lo for lo, hi in bounds) <= 1024
        main = torch.cuda.current_stream()
        streams = self._group_streams(n_micro, dev)
        sess, pos = self.decoder_session, self.positional_embedding
        groups = []
        for g, (lo, hi) in enumerate(bounds):
            groups.append(dict(lo=lo, hi=hi, stream=streams[g].cuda_stream, slot=g, active=True,
                               kv=[t[lo:hi] for t in st['kv']], cross=[t[lo:hi] for t in cross],
                               logits=st['logits'][lo:hi], tokens=st['tokens'][lo:hi], done=st['done'][lo:hi],
                               live=self._live_list(st, n_micro, g, lo, hi) if use_live else None))
        for s_ in streams:
            s_.wait_stream(main)
        cur = L0
        steps_done = 0
        lib = native.load_library()
        use_graph = self.use_graphs and self.decoder_session.qkv_amax is None
        # stream-parallel groups: the steps of two groups are in flight at once, so none of them may take a one-launch form (two 256-workgroup
        # launches side by side can each hold half of the chip and wait for the other half: ADVICE r5, whisper_mi355.h `not_alone`)
        shared = (n_micro > 1 and not self.groups_sequential) or self.force_not_alone

        def finish_step(gr, counter):
            # end of a group's step: advance its device step counter (graph replay) and, with per-row completion,
            # rebuild its list of live rows from the flags the greedy kernel has just updated
            if gr['live'] is not None:
                native.check(lib.wm_step_finish(counter.data_ptr() if counter is not None else None, gr['done'].data_ptr(),
                                                gr['hi'] - gr['lo'], gr['live'].data_ptr(), gr['stream']), "wm_step_finish")
            elif counter is not None:
                native.check(lib.wm_step_advance(counter.data_ptr(), gr['stream']), "wm_step_advance")

        last_issued = None
        for i in range(self.sample_len):
            for gr in groups:
                if not gr['active']:
                    continue
                lo, hi, sm, slot = gr['lo'], gr['hi'], gr['stream'], gr['slot']
                if self.groups_sequential and n_micro > 1:
                    if last_issued is not None:
                        streams[slot].wait_stream(streams[last_issued])
                    last_issued = slot
                gkey = (n_micro, slot, use_live)
                if i == 0:
                    # the prefill (L0 tokens on an empty cache) + the first greedy step.  Round 6: replayed from a graph from the second
                    # batch on -- the call is the same for every batch (state buffers, L0 and the start position are fixed), and issued
                    # eagerly its ~ 390 launches per group are HOST-bound at small and middle batches (7 us each: 2.7 ms per group where the
                    # GPU needs 1.7 at one utterance; with the language pass 9.5 ms per batch of 2 x 8)
                    pkey = (n_micro, slot, use_live, 'prefill', L0, shared)

                    def issue_prefill(gr=gr, lo=lo, hi=hi, sm=sm, slot=slot):
                        sess.decoder_step(gr['tokens'][:, :L0], pos[0:L0], gr['cross'], None, cap, gr['kv'], cap,
                                          gr['logits'], 0, sm, slot=slot, live_rows=gr['live'], not_alone=shared)
                        self._greedy(st, lo, hi, gr['logits'].data_ptr() + (L0 - 1) * V * 2, L0 * V, L0, sm)
                        finish_step(gr, None)
                    if use_graph and self.graph_prefill and pkey in st['graphs']:
                        with torch.cuda.stream(streams[slot]):
                            st['graphs'][pkey].replay()
                    else:
                        issue_prefill()
                        if use_graph and self.graph_prefill and L0 <= 4:        # (longer start sequences run as 4-token passes with copies between them: eager)
                            self._capture(st, pkey, streams[slot], issue_prefill)
                elif use_graph and gkey in st['graphs']:
                    if i == 1 or st['counters'][gkey + ('fresh',)]:
                        # the device step counter holds n_past = cur - 1; (re)seed it on the group's stream
                        with torch.cuda.stream(streams[slot]):
                            st['counters'][gkey].fill_(cur - 1)
                        st['counters'][gkey + ('fresh',)] = False
                    with torch.cuda.stream(streams[slot]):
                        st['graphs'][gkey].replay()
                else:
                    sess.decoder_step(gr['tokens'][:, cur - 1:cur], pos[cur - 1:cur], gr['cross'], gr['kv'], cap,
                                      gr['kv'], cap, gr['logits'], cur - 1, sm, slot=slot, live_rows=gr['live'], not_alone=shared)
                    self._greedy(st, lo, hi, gr['logits'].data_ptr(), V, cur, sm)
                    finish_step(gr, None)
                    if use_graph:
                        # capture ONE decode step (decoder + fused greedy + counter advance) of this group;
                        # every later token replays it: T is read from a device counter inside the kernels
                        counter = torch.zeros(1, dtype=torch.int32, device=dev)
                        streams[slot].synchronize()
                        graph = torch.cuda.CUDAGraph()
                        # capture_error_mode "thread_local" + CAPTURE_LOCK: the encoder of the NEXT batch may be in flight on a helper thread
                        # (WhisperEncoding.prefetch), which waits for events and issues launches on its own stream.  Under the default
                        # "global" mode such a call from ANY thread while this capture is open invalidates it (hipErrorStreamCaptureInvalidated
                        # -- seen once in ~ 25 bench runs: the window is three captures of a millisecond or two against one event wait per
                        # encoder layer); the lock keeps the helper out of the driver for the length of the capture as well.
                        # (torch.cuda.graph's __enter__ synchronises the DEVICE while the lock is held: during a group's first capture the
                        # helper cannot issue its next encoder layer until the layer in flight has finished -- 45-135 ms at B = 576, once
                        # per group and process; the group stream was synchronised just above, so nothing of this loop is waited for)
                        with native.CAPTURE_LOCK, torch.cuda.graph(graph, stream=streams[slot], capture_error_mode="thread_local"):
                            sess.decoder_step(gr['tokens'], pos, gr['cross'], gr['kv'], cap, gr['kv'], cap,
                                              gr['logits'], 1, sm, slot=slot, n_past_dev=counter, n_new=1, live_rows=gr['live'],
                                              not_alone=shared)
                            self._greedy(st, lo, hi, gr['logits'].data_ptr(), V, 0, sm, n_past_dev=counter)
                            finish_step(gr, counter)
                        st['graphs'][gkey], st['counters'][gkey] = graph, counter
                        st['counters'][gkey + ('fresh',)] = True
            if i == 0 and self.tokenizer.no_speech is not None:
                # the greedy kernel has already written -inf into the LAST position's row only;
                # the <|sot|> position (no-speech probability, decoding.py:803-807) is untouched
                for s_ in streams:
                    main.wait_stream(s_)
                probs_at_sot = st['logits'][:, self.sot_index].float().softmax(dim=-1)
                nsp_dev = probs_at_sot[:, self.tokenizer.no_speech]
                for s_ in streams:
                    s_.wait_stream(main)
            if i == 0 and self.first_token_event is not None:
                # latency probe (bench.py): the first sampled token of every row exists once the groups' first steps have run
                for s_ in streams:
                    main.wait_stream(s_)
                self.first_token_event.record(main)
                for s_ in streams:
                    s_.wait_stream(main)
            cur += 1
            steps_done += 1
            if cur > cap:
                break
            if not ignore_eot and (steps_done % self.poll_every == 0):
                # `done` is sticky (a finished row stays at EOT), so "every row is done" is monotone;
                # poll it now and then instead of synchronising every step
                for s_ in streams:
                    main.wait_stream(s_)
                done_host = st['done'].bool().cpu()
                if bool(done_host.all()):
                    break
                if use_live:
                    for gr in groups:         # a group with no live row left costs a whole chain of launches per token: drop it
                        if gr['active'] and bool(done_host[gr['lo']:gr['hi']].all()):
                            gr['active'] = False
        for s_ in streams:
            main.wait_stream(s_)
        out = self._finish_main_loop(st, cur, L0, n_batch, ignore_eot,
                                     nsp_dev if self.tokenizer.no_speech is not None else None)
        self._rep_cache = None        # the candidates' repeated features (best_of): the language pass and this loop have both used them
        if one_row and not _retry and self._chain_gave_up("decode loop"):
            # one-row groups run the token step as ONE launch whose workgroups wait for each other with bounded spins; a wait that was
            # given up invalidates the step and every token after it.  The utterance is decoded again, in this process, on the
            # launch-per-kernel path (the library has taken the device off the one-launch forms; the captured graphs are gone).
            return self.main_loop(features_in, ignore_eot=ignore_eot, row_limit=row_limit, _retry=True)
        return out

    def _main_loop_partitioned(self, audio_features, st, cross, L0, n_micro, bounds, ignore_eot):
        """The decode loop scheduled for the chip (wm_decoder_step_multi): every group's cross-attention kernel
        on one stream that owns most CUs, each group's short kernels on a stream confined to the remaining
        CUs, eager launches with the step counter on the device (the call is identical for every token).
        Same arithmetic, same tokens as the single-stream loop."""
        dev = audio_features.device
        cfg = self.decoder_config
        V, cap = cfg['vocab_size'], cfg['num_text_ctx']
        n_batch = st['tokens'].shape[0]
        sess, pos, lib = self.decoder_session, self.positional_embedding, native.load_library()
        main = torch.cuda.current_stream()
        light, heavy = self._partition_streams(n_micro, dev)
        for s_ in light + [heavy]:
            s_.wait_stream(main)
        light_ptrs, heavy_ptr = [s_.cuda_stream for s_ in light], heavy.cuda_stream
        key = ('partition', n_micro)
        if key not in st['graphs']:
            counters = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in bounds]
            views = [dict(kv=[t[lo:hi] for t in st['kv']], cross=[t[lo:hi] for t in cross], logits=st['logits'][lo:hi],
                          tokens=st['tokens'][lo:hi]) for lo, hi in bounds]
            prefill = [sess.make_decoder_io(v['tokens'][:, :L0], pos[0:L0], v['cross'], None, cap, v['kv'], cap, v['logits'], 0,
                                            slot=g) for g, v in enumerate(views)]
            step = [sess.make_decoder_io(v['tokens'], pos, v['cross'], v['kv'], cap, v['kv'], cap, v['logits'], 1, slot=g,
                                         n_past_dev=counters[g], n_new=1) for g, v in enumerate(views)]
            st['graphs'][key] = dict(counters=counters, prefill=prefill, step=step, views=views)
        plan = st['graphs'][key]
        cur = L0
        steps_done = 0
        in_flight = []            # one event per issued step: the three queues only interleave well while the host
        for i in range(self.sample_len):          # is a few steps ahead, not when it runs into a full queue mid-step
            if len(in_flight) >= self.run_ahead:
                in_flight.pop(0).synchronize()
            if i == 0:
                sess.decoder_step_multi(plan['prefill'], light_ptrs, heavy_ptr)
                for g, (lo, hi) in enumerate(bounds):
                    self._greedy(st, lo, hi, plan['views'][g]['logits'].data_ptr() + (L0 - 1) * V * 2, L0 * V, cur, light_ptrs[g])
                if self.tokenizer.no_speech is not None:
                    # the greedy kernel wrote -inf into the LAST position's row only; the <|sot|> row is untouched
                    for s_ in light:
                        main.wait_stream(s_)
                    nsp_dev = st['logits'][:, self.sot_index].float().softmax(dim=-1)[:, self.tokenizer.no_speech]
                    for s_ in light:
                        s_.wait_stream(main)
            else:
                if i == 1:
                    for g, s_ in enumerate(light):
                        with torch.cuda.stream(s_):
                            plan['counters'][g].fill_(cur - 1)      # the device step counter holds n_past = cur - 1
                sess.decoder_step_multi(plan['step'], light_ptrs, heavy_ptr)
                for g, (lo, hi) in enumerate(bounds):
                    self._greedy(st, lo, hi, plan['views'][g]['logits'].data_ptr(), V, 0, light_ptrs[g],
                                 n_past_dev=plan['counters'][g])
                    native.check(lib.wm_step_advance(plan['counters'][g].data_ptr(), light_ptrs[g]), "wm_step_advance")
            done = torch.cuda.Event()
            done.record(light[0])
            in_flight.append(done)
            cur += 1
            steps_done += 1
            if cur > cap:
                break
            if not ignore_eot and (steps_done % self.poll_every == 0):
                for s_ in light:
                    main.wait_stream(s_)
                if bool((st['tokens'][:, cur - 1] == self.tokenizer.eot).all()):
                    break
        for s_ in light + [heavy]:
            main.wait_stream(s_)
        return self._finish_main_loop(st, cur, L0, n_batch, ignore_eot,
                                      nsp_dev if self.tokenizer.no_speech is not None else None)

    def _finish_main_loop(self, st, cur, L0, n_batch, ignore_eot, nsp_dev):
        tokens = st['tokens'][:, :cur].to(torch.int64)
        if not ignore_eot:
            # a finished row stays at EOT (W/decoding.py:295); the columns a dropped group no longer wrote are EOT too
            eot = self.tokenizer.eot
            after = (tokens[:, L0:] == eot).cumsum(dim=1) > 0
            tokens[:, L0:] = torch.where(after, torch.full_like(tokens[:, L0:], eot), tokens[:, L0:])
            # cut at the first column where every row is EOT: where the per-step check would have stopped
            all_eot = (tokens[:, L0:] == self.tokenizer.eot).all(dim=0)
            if bool(all_eot.any()):
                tokens = tokens[:, :L0 + int(all_eot.float().argmax()) + 1]
        no_speech_probs = [np.nan] * n_batch
        if nsp_dev is not None:
            no_speech_probs = nsp_dev.tolist()
        return tokens, st['sum_logprobs'].clone(), no_speech_probs

    # ---- post-processing -------------------------------------------------------------------------------
    def compression_ratio(self, text) -> float:
        text_bytes = text.encode("utf-8")
        return len(text_bytes) / len(zlib.compress(text_bytes))

    def post_process(self, tokens, sum_logprobs, no_speech_probs, audio_features, languages):
        """Slice, rank, detokenise (W/decoding.py:827-878); n_audio comes from the batch, not the config."""
        if audio_features.shape[0] == len(no_speech_probs):          # already one row per candidate (the reference's convention)
            audio_features = audio_features[:: self.n_group]
        no_speech_probs = no_speech_probs[:: self.n_group]
        n_audio = audio_features.shape[0]
        assert n_audio == len(no_speech_probs)
        tokens = tokens.reshape(n_audio, self.n_group, -1)
        sum_logprobs = sum_logprobs.reshape(n_audio, self.n_group)
        tokens, sum_logprobs = self.decoder.finalize(tokens, sum_logprobs)
        eot = self.tokenizer.eot
        tokens = [[t[self.sample_begin: (t == eot).nonzero()[0, 0]] for t in s] for s in tokens]
        selected = self.sequence_ranker.rank(tokens, sum_logprobs)
        tokens = [t[i].tolist() for i, t in zip(selected, tokens)]
        texts = [self.tokenizer.decode(t).strip() for t in tokens]
        sum_logprobs = [lp[i] for i, lp in zip(selected, sum_logprobs)]
        avg_logprobs = [lp / (len(t) + 1) for t, lp in zip(tokens, sum_logprobs)]
        fields = (texts, languages, tokens, audio_features, avg_logprobs, no_speech_probs)
        if len(set(map(len, fields))) != 1:
            raise RuntimeError(f"inconsistent result lengths: {list(map(len, fields))}")
        return [
            DecodingResult(audio_features=features, language=language, tokens=toks, text=text, avg_logprob=avg,
                           no_speech_prob=nsp, temperature=self.options.temperature,
                           compression_ratio=self.compression_ratio(text))
            for text, language, toks, features, avg, nsp in zip(*fields)
        ]
