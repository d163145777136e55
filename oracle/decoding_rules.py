"""CPU oracle for Whisper's greedy decoding rules  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, row by row and without the reference's classes, what W/decoding.py applies to the
last-position logits of every step (W/ = /root/reference/tensorrt_llm_july-release-v1/examples/whisper):

    SuppressBlank          W/decoding.py:202-209
    SuppressTokens         W/decoding.py:212-217   (list built by _get_suppress_tokens :394-421)
    ApplyTimestampRules    W/decoding.py:134-199
    GreedyDecoder.update   W/decoding.py:274-300   (temperature 0 restated; temperature > 0 pinned through goldens)
    MaximumLikelihoodRanker W/decoding.py:92-115
    detect_language        W/decoding.py:703-741   (mask to language tokens, argmax, softmax)
    no_speech_prob         W/decoding.py:803-807
    main_loop              W/decoding.py:785-821

Pinned by tests/golden/decoding_rules.npz, which oracle/gen_golden.py produced by running the
reference's own classes (imported from /root/reference with stub modules for tiktoken/tensorrt)
on seeded logits and token histories, and by tests/golden/sampling.npz (same generator): the
reference's GreedyDecoder at temperature 0.7, its MaximumLikelihoodRanker with and without a
length penalty, and its main_loop + post_process driven with best_of = 3 on the seeded logits of
`sampling_logits` below.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

NEG_INF = -np.inf


@dataclass(frozen=True)
class SpecialIds:
    """Special-token ids as laid out by W/decoding.py:433-449: the specials follow the BPE ranks
    in the order eot, sot, 99 languages, translate, transcribe, startoflm, startofprev,
    nospeech, notimestamps, then 1501 timestamps."""
    n_base: int          # number of BPE ranks (50257 multilingual, 50256 gpt2)
    n_langs: int = 99

    @property
    def eot(self): return self.n_base
    @property
    def sot(self): return self.n_base + 1
    @property
    def lang0(self): return self.n_base + 2
    @property
    def translate(self): return self.n_base + 2 + self.n_langs
    @property
    def transcribe(self): return self.translate + 1
    @property
    def sot_lm(self): return self.translate + 2
    @property
    def sot_prev(self): return self.translate + 3
    @property
    def no_speech(self): return self.translate + 4
    @property
    def no_timestamps(self): return self.translate + 5
    @property
    def timestamp_begin(self): return self.translate + 6
    @property
    def n_vocab(self): return self.timestamp_begin + 1501


MULTILINGUAL = SpecialIds(50257)


@dataclass
class RuleSet:
    ids: SpecialIds
    sample_begin: int                       # len(sot_sequence) = 3 (W/decoding.py:340)
    suppress_tokens: Sequence[int]          # sorted, incl. the specials added at :407-419
    blank_tokens: Sequence[int]             # tokenizer.encode(" ") + [eot]   (:209)
    max_initial_timestamp_index: Optional[int] = 50   # round(1.0 / 0.02)   (:343-348)
    timestamps: bool = True                 # False = DecodingOptions.without_timestamps: no ApplyTimestampRules filter is built (:337-348)


def log_softmax_f32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    m = x.max(axis=-1, keepdims=True)
    with np.errstate(divide="ignore"):
        return (x - m) - np.log(np.exp(x - m).sum(axis=-1, keepdims=True, dtype=np.float32))


def apply_filters(logits: np.ndarray, tokens: np.ndarray, rules: RuleSet, dominance_out: Optional[list] = None) -> np.ndarray:
    """logits [B, V] (modified copy is returned), tokens [B, cur_len] = the whole context so far.
    `dominance_out` (tests): receives, per row, log P(any timestamp) - max log P(text token), the quantity whose sign
    decides the last rule -- a near-zero value is a near-tie of that rule, not of two logits."""
    lg = logits.astype(np.float32).copy()
    ids = rules.ids
    B, cur = tokens.shape
    tb = ids.timestamp_begin
    # SuppressBlank (:202-209)
    if cur == rules.sample_begin:
        lg[:, list(rules.blank_tokens)] = NEG_INF
    # SuppressTokens (:212-217)
    lg[:, list(rules.suppress_tokens)] = NEG_INF
    if not rules.timestamps:                # without_timestamps: the two suppress filters are the whole list (:332-348)
        return lg
    # ApplyTimestampRules (:145-199)
    lg[:, ids.no_timestamps] = NEG_INF
    for k in range(B):
        seq = tokens[k, rules.sample_begin:]
        last_ts = len(seq) >= 1 and seq[-1] >= tb
        pen_ts = len(seq) < 2 or seq[-2] >= tb
        if last_ts:
            if pen_ts:
                lg[k, tb:] = NEG_INF
            else:
                lg[k, :ids.eot] = NEG_INF
        ts = seq[seq >= tb]
        if ts.size > 0:
            if last_ts and not pen_ts:
                ts_last = int(ts[-1])
            else:
                ts_last = int(ts[-1]) + 1
            lg[k, tb:ts_last] = NEG_INF
    if cur == rules.sample_begin:
        lg[:, :tb] = NEG_INF
        if rules.max_initial_timestamp_index is not None:
            lg[:, tb + rules.max_initial_timestamp_index + 1:] = NEG_INF
    lp = log_softmax_f32(lg)
    for k in range(B):
        row = lp[k, tb:]
        m = row.max()
        with np.errstate(divide="ignore", invalid="ignore"):
            ts_lp = m + np.log(np.exp(row - m).sum(dtype=np.float32)) if np.isfinite(m) else NEG_INF
        if dominance_out is not None:
            dominance_out.append(float(ts_lp - lp[k, :tb].max()))
        if ts_lp > lp[k, :tb].max():
            lg[k, :tb] = NEG_INF
    return lg


def greedy_update(tokens: np.ndarray, logits: np.ndarray, sum_logprobs: np.ndarray, eot: int
                  ) -> Tuple[np.ndarray, bool]:
    """GreedyDecoder.update at temperature 0 (:278-300).  sum_logprobs updated in place."""
    nxt = logits.argmax(axis=-1)
    lp = log_softmax_f32(logits)
    cur = lp[np.arange(lp.shape[0]), nxt]
    alive = tokens[:, -1] != eot
    sum_logprobs += np.where(alive, cur, np.float32(0)).astype(np.float32)
    nxt = np.where(alive, nxt, eot)
    tokens = np.concatenate([tokens, nxt[:, None]], axis=1)
    return tokens, bool((tokens[:, -1] == eot).all())


def detect_language(logits_sot: np.ndarray, ids: SpecialIds):
    """logits at the single <|sot|> position [B, V] -> (language token ids, probs over the 99)."""
    lg = logits_sot.astype(np.float32).copy()
    mask = np.ones(lg.shape[-1], dtype=bool)
    mask[ids.lang0:ids.lang0 + ids.n_langs] = False
    lg[:, mask] = NEG_INF
    tok = lg.argmax(axis=-1)
    p = np.exp(log_softmax_f32(lg))[:, ids.lang0:ids.lang0 + ids.n_langs]
    return tok, p


def no_speech_prob(logits_at_sot: np.ndarray, ids: SpecialIds) -> np.ndarray:
    return np.exp(log_softmax_f32(logits_at_sot))[:, ids.no_speech]


def main_loop(step_fn, init_tokens: np.ndarray, rules: RuleSet, sample_len: int, n_text_ctx: int,
              ignore_eot: bool = False):
    """W/decoding.py:785-821 with the model abstracted as
    `step_fn(tokens_this_step [B,L], is_first) -> logits [B, L, V]` (keeps its own KV cache)."""
    tokens = init_tokens.copy()
    B = tokens.shape[0]
    sum_lp = np.zeros(B, dtype=np.float32)
    nsp = [float("nan")] * B
    for i in range(sample_len):
        feed = tokens if i == 0 else tokens[:, -1:]
        logits = step_fn(feed, i == 0)
        if i == 0:
            sot_index = list(init_tokens[0]).index(rules.ids.sot)
            nsp = no_speech_prob(logits[:, sot_index], rules.ids).tolist()
        lg = apply_filters(logits[:, -1], tokens, rules)
        tokens, done = greedy_update(tokens, lg, sum_lp, rules.ids.eot)
        if (done and not ignore_eot) or tokens.shape[1] > n_text_ctx:
            break
    return tokens, sum_lp, nsp


def golden_rule_cases(ids: SpecialIds = MULTILINGUAL):
    """The (token history, logits) inputs of tests/golden/decoding_rules.npz, regenerated from a
    Philox stream so the fixture stores expected OUTPUTS only.  Logits are fp16-representable
    (the engine's logits dtype) and returned as fp32."""
    rng = np.random.Generator(np.random.Philox(5))
    tb, V = ids.timestamp_begin, ids.n_vocab
    sot_seq = [ids.sot, ids.lang0, ids.transcribe]
    histories = [
        [], [tb + 3], [tb + 3, 400], [tb + 3, 400, 500], [tb + 3, 400, tb + 40],
        [tb + 3, 400, tb + 40, tb + 40], [tb + 3, 400, tb + 40, tb + 40, 321],
        [tb + 0, 11, 12, tb + 100, tb + 100, 13, tb + 1500], [tb + 10, 7, ids.eot],
        [tb + 10, 7, ids.eot, ids.eot], [tb + 1500], [tb + 5, 220, 50256, tb + 1499, tb + 1499],
    ]
    cases = []
    for hist in histories:
        for variant in range(3):
            logits = (rng.standard_normal(V) * 2.0).astype(np.float32)
            if variant == 1:      # timestamps collectively likely
                logits[tb:] += 4.0
            if variant == 2:      # one dominant text token, one dominant timestamp
                logits[int(rng.integers(0, 50000))] += 12.0
                logits[tb + int(rng.integers(0, 1501))] += 9.0
            cases.append((np.array(sot_seq + hist, dtype=np.int64),
                          logits.astype(np.float16).astype(np.float32)))
    return cases


def rank_max_likelihood(lengths: Sequence[Sequence[int]], sum_logprobs: Sequence[Sequence[float]],
                        length_penalty: Optional[float]) -> List[int]:
    """MaximumLikelihoodRanker.rank (W/decoding.py:92-115): per group, the index of the highest
    sum_logprob / penalty, penalty = length, or ((5 + length) / 6) ** length_penalty (Google NMT)."""
    picks = []
    for ls, lps in zip(lengths, sum_logprobs):
        sc = [lp / (n if length_penalty is None else ((5 + n) / 6) ** length_penalty) for n, lp in zip(ls, lps)]
        picks.append(int(np.argmax(sc)))
    return picks


def sampling_logits(step: int, n_rows: int, n_tokens: int, ids: SpecialIds = MULTILINGUAL, seed: int = 4242) -> np.ndarray:
    """Seeded decoder outputs [n_rows, n_tokens, V] (fp32 values that are fp16-representable) for the sampling
    goldens: the stand-in for `decode()` that both the reference's main_loop (in oracle/gen_golden.py) and the
    product's main_loop_reference (in tests/) are driven with.  A few text tokens and timestamps stand out so
    that temperature 0.7 picks among a handful of candidates; from step 3 on EOT gains weight row by row, so
    candidates of one utterance end at different lengths (what the ranker's length term needs)."""
    rng = np.random.Generator(np.random.Philox(key=seed + 1000 * step))
    V, tb = ids.n_vocab, ids.timestamp_begin
    lg = (rng.standard_normal((n_rows, n_tokens, V)) * 1.5).astype(np.float32)
    for r in range(n_rows):
        hot = rng.integers(300, 40000, size=6)
        lg[r, :, hot] += rng.uniform(7.0, 10.0, size=6).astype(np.float32)[:, None]
        ts = tb + rng.integers(0, 1400, size=3)
        lg[r, :, ts] += rng.uniform(5.0, 9.0, size=3).astype(np.float32)[:, None]
        if step >= 3:
            lg[r, :, ids.eot] += np.float32(6.0 + 1.5 * ((r + step) % 4))
    return lg.astype(np.float16).astype(np.float32)
