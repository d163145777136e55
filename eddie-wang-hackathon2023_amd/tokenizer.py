"""Whisper tokenizer for the engine: special-token ids, the suppress list and (when a vocabulary
file is supplied) BPE encode/decode.

Mirrors the surface `WhisperDecoding` uses from the reference tokenizer
(/root/reference/tensorrt_llm_july-release-v1/examples/whisper/tokenizer.py:125-251 and
decoding.py:423-481): `sot_sequence`, `eot`, `sot`, `transcribe`, `translate`, `sot_lm`,
`sot_prev`, `no_speech`, `no_timestamps`, `timestamp_begin`, `all_language_tokens`,
`all_language_codes`, `non_speech_tokens`, `encode`, `decode`, `decode_with_timestamps`.

The reference delegates BPE to the `tiktoken` wheel (absent here, no network).  This file
implements byte-pair merging itself, in the published tiktoken order: split with the GPT-2
pattern, then repeatedly merge the adjacent pair with the lowest rank.

Two modes:
* vocabulary mode  -- `Tokenizer.from_vocab(path_to_*.tiktoken, ...)`: everything works.
* ids-only mode    -- `Tokenizer.ids_only(multilingual=True)`: no vocabulary file on the box
  (the GPU box has none); token ids, the special layout and the suppress list (constants below,
  derived with the vocabulary mode and frozen in tests/golden/tokenizer.npz) work, `decode`
  renders `<|id|>` placeholders and `encode` raises.
"""
from __future__ import annotations

import base64
from dataclasses import dataclass, field
from functools import cached_property
from typing import Dict, List, Optional, Sequence, Tuple

# ISO codes in Whisper's language-token order (data; order defines ids sot+1 ... sot+99)
LANGUAGE_CODES: Tuple[str, ...] = tuple(
    "en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no "
    "th ur hr bg lt la mi ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw "
    "gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl "
    "mg as tt haw ln ha ba jw su".split())
LANGUAGE_NAMES: Tuple[str, ...] = tuple(
    "english chinese german spanish russian korean french japanese portuguese turkish polish "
    "catalan dutch arabic swedish italian indonesian hindi finnish vietnamese hebrew ukrainian "
    "greek malay czech romanian danish hungarian tamil norwegian thai urdu croatian bulgarian "
    "lithuanian latin maori malayalam welsh slovak telugu persian latvian bengali serbian "
    "azerbaijani slovenian kannada estonian macedonian breton basque icelandic armenian nepali "
    "mongolian bosnian kazakh albanian swahili galician marathi punjabi sinhala khmer shona "
    "yoruba somali afrikaans occitan georgian belarusian tajik sindhi gujarati amharic yiddish "
    "lao uzbek faroese haitian_creole pashto turkmen nynorsk maltese sanskrit luxembourgish "
    "myanmar tibetan tagalog malagasy assamese tatar hawaiian lingala hausa bashkir javanese "
    "sundanese".split())
LANGUAGES: Dict[str, str] = {c: n.replace("_", " ") for c, n in zip(LANGUAGE_CODES, LANGUAGE_NAMES)}
TO_LANGUAGE_CODE: Dict[str, str] = {
    **{n: c for c, n in LANGUAGES.items()},
    "burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb",
    "pushto": "ps", "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si",
    "castilian": "es",
}
assert len(LANGUAGE_CODES) == len(LANGUAGE_NAMES) == 99

GPT2_SPLIT = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""

# Frozen outputs of the vocabulary mode on the multilingual vocabulary (see tests/golden/tokenizer.npz
# and tests/test_tokenizer.py): tokenizer.non_speech_tokens and tokenizer.encode(" ").
MULTILINGUAL_NON_SPEECH: Tuple[int, ...] = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359,
    503, 522, 542, 873, 893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268,
    3536, 3846, 3961, 4183, 4667, 6585, 6647, 7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331,
    12562, 13793, 14157, 14635, 15265, 15618, 16553, 16604, 18362, 18956, 20075, 21675, 22520,
    26130, 26161, 26435, 28279, 29464, 31650, 32302, 32470, 36865, 42863, 47425, 49870, 50254)
GPT2_NON_SPEECH: Tuple[int, ...] = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 357,
    366, 438, 532, 685, 705, 796, 930, 1058, 1220, 1267, 1279, 1303, 1343, 1377, 1391, 1635, 1782,
    1875, 2162, 2361, 2488, 3467, 4008, 4211, 4600, 4808, 5299, 5855, 6329, 7203, 9609, 9959,
    10563, 10786, 11420, 11709, 11907, 13163, 13697, 13700, 14808, 15306, 16410, 16791, 17992,
    19203, 19510, 20724, 22305, 22935, 27007, 30109, 30420, 33409, 34949, 40283, 40493, 40549,
    47282, 49146)
BLANK_TOKENS: Tuple[int, ...] = (220,)      # encode(" ") in both vocabularies


class BPE:
    """Byte-pair encoder over a `token_bytes -> rank` table (the *.tiktoken format: one
    `base64(token) rank` pair per line, W/decoding.py:425-429)."""

    def __init__(self, ranks: Dict[bytes, int]):
        import regex  # third-party `regex` (for \p{L}); present in the image
        self.ranks = ranks
        self.by_rank = {r: b for b, r in ranks.items()}
        self._split = regex.compile(GPT2_SPLIT)

    @classmethod
    def from_file(cls, path: str) -> "BPE":
        ranks = {}
        with open(path) as f:
            for line in f:
                if line.strip():
                    tok, rank = line.split()
                    ranks[base64.b64decode(tok)] = int(rank)
        return cls(ranks)

    def _merge(self, piece: bytes) -> List[int]:
        if piece in self.ranks:
            return [self.ranks[piece]]
        parts = [piece[i:i + 1] for i in range(len(piece))]
        while len(parts) > 1:
            best, best_i = None, -1
            for i in range(len(parts) - 1):
                r = self.ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best is None or r < best):
                    best, best_i = r, i
            if best is None:
                break
            parts[best_i:best_i + 2] = [parts[best_i] + parts[best_i + 1]]
        return [self.ranks[p] for p in parts]

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        for m in self._split.findall(text):
            out.extend(self._merge(m.encode("utf-8")))
        return out

    def decode_bytes(self, ids: Sequence[int]) -> bytes:
        return b"".join(self.by_rank[i] for i in ids)


def special_token_table(n_base: int) -> Dict[str, int]:
    """Specials appended after the BPE ranks in the reference's order (W/decoding.py:433-449)."""
    names = ["<|endoftext|>", "<|startoftranscript|>", *[f"<|{c}|>" for c in LANGUAGE_CODES],
             "<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>",
             "<|nospeech|>", "<|notimestamps|>", *[f"<|{i * 0.02:.2f}|>" for i in range(1501)]]
    return {n: n_base + i for i, n in enumerate(names)}


@dataclass
class Tokenizer:
    n_base: int                                  # 50257 multilingual / 50256 gpt2
    language: Optional[str] = None
    task: Optional[str] = None
    bpe: Optional[BPE] = None
    special_tokens: Dict[str, int] = field(default_factory=dict)
    sot_sequence: Tuple[int, ...] = ()

    def __post_init__(self):
        self.special_tokens = special_token_table(self.n_base)
        self._special_by_id = {v: k for k, v in self.special_tokens.items()}
        seq = [self.sot]
        if self.language is not None:
            seq.append(self.sot + 1 + LANGUAGE_CODES.index(self.language))
        if self.task is not None:
            seq.append(self.transcribe if self.task == "transcribe" else self.translate)
        self.sot_sequence = tuple(seq)

    # -- constructors ------------------------------------------------------------------
    @classmethod
    def from_vocab(cls, path: str, multilingual: bool = True, language: Optional[str] = None,
                   task: Optional[str] = None) -> "Tokenizer":
        bpe = BPE.from_file(path)
        language, task = cls._defaults(multilingual, language, task)
        return cls(n_base=len(bpe.ranks), language=language, task=task, bpe=bpe)

    @classmethod
    def ids_only(cls, multilingual: bool = True, language: Optional[str] = None,
                 task: Optional[str] = None) -> "Tokenizer":
        language, task = cls._defaults(multilingual, language, task)
        return cls(n_base=50257 if multilingual else 50256, language=language, task=task)

    @staticmethod
    def _defaults(multilingual, language, task):
        # W/decoding.py:458-481
        if language is not None:
            language = language.lower()
            if language not in LANGUAGES:
                if language in TO_LANGUAGE_CODE:
                    language = TO_LANGUAGE_CODE[language]
                else:
                    raise ValueError(f"Unsupported language: {language}")
        if multilingual:
            return language or "en", task or "transcribe"
        return None, None

    # -- ids ---------------------------------------------------------------------------
    @property
    def n_vocab(self) -> int: return self.n_base + len(self.special_tokens)
    @property
    def eot(self) -> int: return self.special_tokens["<|endoftext|>"]
    @property
    def sot(self) -> int: return self.special_tokens["<|startoftranscript|>"]
    @property
    def translate(self) -> int: return self.special_tokens["<|translate|>"]
    @property
    def transcribe(self) -> int: return self.special_tokens["<|transcribe|>"]
    @property
    def sot_lm(self) -> int: return self.special_tokens["<|startoflm|>"]
    @property
    def sot_prev(self) -> int: return self.special_tokens["<|startofprev|>"]
    @property
    def no_speech(self) -> int: return self.special_tokens["<|nospeech|>"]
    @property
    def no_timestamps(self) -> int: return self.special_tokens["<|notimestamps|>"]
    @property
    def timestamp_begin(self) -> int: return self.special_tokens["<|0.00|>"]

    @property
    def language_token(self) -> int:
        if self.language is None:
            raise ValueError("This tokenizer does not have language token configured")
        return self.special_tokens[f"<|{self.language}|>"]

    @cached_property
    def all_language_tokens(self) -> Tuple[int, ...]:
        return tuple(self.special_tokens[f"<|{c}|>"] for c in LANGUAGE_CODES)

    @cached_property
    def all_language_codes(self) -> Tuple[str, ...]:
        return LANGUAGE_CODES

    @cached_property
    def sot_sequence_including_notimestamps(self) -> Tuple[int, ...]:
        return tuple(list(self.sot_sequence) + [self.no_timestamps])

    # -- text --------------------------------------------------------------------------
    def encode(self, text: str) -> List[int]:
        if self.bpe is None:
            raise RuntimeError("tokenizer is in ids-only mode: no vocabulary file was supplied")
        return self.bpe.encode(text)

    def blank_tokens(self) -> Tuple[int, ...]:
        """tokenizer.encode(" ") as SuppressBlank needs it (W/decoding.py:209)."""
        if self.bpe is not None:
            return tuple(self.bpe.encode(" "))
        return BLANK_TOKENS

    def decode(self, token_ids: Sequence[int]) -> str:
        return self.decode_with_timestamps([t for t in token_ids if t < self.timestamp_begin])

    def decode_with_timestamps(self, token_ids: Sequence[int]) -> str:
        out, run = [], []

        def flush():
            if run:
                if self.bpe is not None:
                    out.append(self.bpe.decode_bytes(run).decode("utf-8", errors="replace"))
                else:
                    out.extend(f"<|{t}|>" for t in run)
                run.clear()

        for t in token_ids:
            t = int(t)
            if t >= self.n_base:
                flush()
                out.append(self._special_by_id[t])
            else:
                run.append(t)
        flush()
        return "".join(out)

    @cached_property
    def non_speech_tokens(self) -> Tuple[int, ...]:
        """Tokens for speaker tags / non-speech annotations (reference tokenizer.py:215-251):
        every listed symbol that is a single token with or without a leading space, the first
        token of each musical-note symbol, plus " -" and " '"."""
        if self.bpe is None:
            return MULTILINGUAL_NON_SPEECH if self.n_base == 50257 else GPT2_NON_SPEECH
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        notes = set("♩♪♫♬♭♮♯")
        result = {self.bpe.encode(" -")[0], self.bpe.encode(" '")[0]}
        for s in symbols + list(notes):
            for toks in (self.bpe.encode(s), self.bpe.encode(" " + s)):
                if len(toks) == 1 or s in notes:
                    result.add(toks[0])
        return tuple(sorted(result))
