"""English transcript normalisation for WER (behaviour of W/normalizers/english.py:1-550).

Three stages, applied by EnglishTextNormalizer in this order:
  1. regex clean-up: bracketed asides, fillers, contractions, titles (english.py:455-550);
  2. EnglishNumberNormalizer: spelled-out numbers -> digits (english.py:12-440).  Implemented here as a
     word classifier (one lexicon lookup per word) feeding a small accumulator machine: the machine holds
     the number being built either as an int (arithmetic composition: "two hundred and five" -> 205) or
     as a digit string (nominal composition: "one oh one" -> 101, "nineteen eighty" -> 1980) plus a
     pending sign / currency prefix;
  3. EnglishSpellingNormalizer: British -> American spellings from english.json (english.py:443-455).
The observable behaviour (including its quirks, e.g. a dangling "point" being dropped) follows the
reference; tests/golden/normalizer.json pins it against the reference's own output.
"""
import json
import os
import re
from fractions import Fraction
from typing import Dict, Iterable, List, Optional, Tuple, Union

from .basic import remove_symbols_and_diacritics

_NUMERIC = re.compile(r"^\d+(\.\d+)?$")

_UNITS = ("one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen "
          "seventeen eighteen nineteen").split()
_TENS = "twenty thirty forty fifty sixty seventy eighty ninety".split()
_SCALES = ("thousand million billion trillion quadrillion quintillion sextillion septillion octillion nonillion "
           "decillion").split()
_IRREGULAR_ORDINALS = {"zeroth": 0, "first": 1, "second": 2, "third": 3, "fifth": 5, "twelfth": 12}
_ORDINAL_ENDINGS = {1: "st", 2: "nd", 3: "rd"}

Number = Union[int, str]


def _as_fraction(v: Number) -> Optional[Fraction]:
    try:
        return Fraction(v)
    except ValueError:
        return None


class EnglishNumberNormalizer:
    """Spelled-out numbers -> arabic digits: drops commas, keeps suffixes (`1960s`, `274th`), moves currency
    words in front as symbols (`twenty dollars` -> `$20`), reads runs of single digits as one nominal number
    (`one oh one` -> `101`), and leaves a lone `one` / `ones` spelled out."""

    def __init__(self):
        self.zeros = {"o", "oh", "zero"}
        self.ones: Dict[str, int] = {w: i + 1 for i, w in enumerate(_UNITS)}
        self.tens: Dict[str, int] = {w: 10 * (i + 2) for i, w in enumerate(_TENS)}
        self.multipliers: Dict[str, int] = {"hundred": 100}
        self.multipliers.update({w: 1000 ** (i + 1) for i, w in enumerate(_SCALES)})

        # word -> (value, suffix) for plurals ("sixes", "twenties", "millions") and ordinals
        self.ones_suffixed: Dict[str, Tuple[int, str]] = {}
        for w, v in self.ones.items():
            self.ones_suffixed[w + ("es" if w == "six" else "s")] = (v, "s")
        for w, v in _IRREGULAR_ORDINALS.items():
            self.ones_suffixed[w] = (v, _ORDINAL_ENDINGS.get(v, "th"))
        for w, v in self.ones.items():
            if v > 3 and v not in (5, 12):
                self.ones_suffixed[w + ("h" if w.endswith("t") else "th")] = (v, "th")
        self.tens_suffixed: Dict[str, Tuple[int, str]] = {}
        for w, v in self.tens.items():
            self.tens_suffixed[w.replace("y", "ies")] = (v, "s")
            self.tens_suffixed[w.replace("y", "ieth")] = (v, "th")
        self.multipliers_suffixed: Dict[str, Tuple[int, str]] = {}
        for w, v in self.multipliers.items():
            self.multipliers_suffixed[w + "s"] = (v, "s")
            self.multipliers_suffixed[w + "th"] = (v, "th")

        self.decimals = set(self.ones) | set(self.tens) | self.zeros
        self.preceding_prefixers = {"minus": "-", "negative": "-", "plus": "+", "positive": "+"}
        self.following_prefixers = {"pound": "£", "pounds": "£", "euro": "€", "euros": "€", "dollar": "$",
                                    "dollars": "$", "cent": "¢", "cents": "¢"}
        self.prefixes = set(self.preceding_prefixers.values()) | set(self.following_prefixers.values())
        self.suffixers = {"per": {"cent": "%"}, "percent": "%"}
        self.specials = {"and", "double", "triple", "point"}

        # one lookup per word: kind + payload
        self._lexicon: Dict[str, Tuple[str, object]] = {}
        for kind, table in (("special", dict.fromkeys(self.specials)), ("percent", self.suffixers),
                            ("currency", self.following_prefixers), ("sign", self.preceding_prefixers),
                            ("scale_sfx", self.multipliers_suffixed), ("scale", self.multipliers),
                            ("ten_sfx", self.tens_suffixed), ("ten", self.tens), ("unit_sfx", self.ones_suffixed),
                            ("unit", self.ones), ("zero", dict.fromkeys(self.zeros))):
            for w, payload in table.items():
                self._lexicon[w] = (kind, payload)          # later (= earlier-tested in the reference) kinds win
        self.words = set(self._lexicon)
        self.literal_words = {"one", "ones"}

    # ---- the accumulator machine ---------------------------------------------------------------------
    def process_words(self, words: List[str]) -> Iterable[str]:
        out: List[str] = []
        value: Optional[Number] = None
        prefix: Optional[str] = None

        def flush(text: Number) -> None:
            """Emit `text` with the pending prefix; the machine returns to the idle state."""
            nonlocal value, prefix
            text = str(text)
            out.append(text if prefix is None else prefix + text)
            value, prefix = None, None

        def flush_pending() -> None:
            if value is not None:
                flush(value)

        def digits(v: Optional[Number]) -> str:
            return str(v or "")

        def scaled(v: Number, m: int) -> Optional[int]:
            """v * m if that is a whole number (v may be a digit string such as '2.5')."""
            f = _as_fraction(v)
            if f is None:
                return None
            p = f * m
            return p.numerator if p.denominator == 1 else None

        def attach_small(n: int, after_ten_word: bool, after_unit_word: bool) -> Number:
            """Fold a 1..19 word into the number so far: arithmetic when a slot is free, else concatenation."""
            if value is None:
                return n
            if isinstance(value, str) or after_unit_word:
                if after_ten_word and n < 10:
                    assert value[-1] == "0"
                    return value[:-1] + str(n)              # "twenty" "one" in a digit string: 20 -> 21
                return str(value) + str(n)
            slot = 10 if n < 10 else 100
            return value + n if value % slot == 0 else str(value) + str(n)

        def attach_ten(t: int) -> Number:
            if value is None:
                return t
            if isinstance(value, str):
                return value + str(t)
            return value + t if value % 100 == 0 else str(value) + str(t)

        def fold_scale(m: int) -> int:
            """int value: the part below one thousand is what the multiplier applies to."""
            return value // 1000 * 1000 + value % 1000 * m

        n_words = len(words)
        skip_next = False
        for i, cur in enumerate(words):
            if skip_next:
                skip_next = False
                continue
            prev = words[i - 1] if i > 0 else None
            nxt = words[i + 1] if i + 1 < n_words else None
            nxt_numeric = nxt is not None and _NUMERIC.match(nxt) is not None
            nxt_numberish = nxt in self.words or nxt_numeric

            signed = cur[0] in self.prefixes
            bare = cur[1:] if signed else cur
            if _NUMERIC.match(bare):                         # arabic digits, possibly signed / decimal
                frac = _as_fraction(bare)
                assert frac is not None
                if value is not None:
                    if isinstance(value, str) and value.endswith("."):
                        value = str(value) + str(cur)        # decimals / dotted components keep concatenating
                        continue
                    flush(value)
                if signed:
                    prefix = cur[0]
                value = frac.numerator if frac.denominator == 1 else bare
                continue

            kind, payload = self._lexicon.get(cur, ("word", None))
            if kind == "word":
                flush_pending()
                flush(cur)
            elif kind == "zero":
                value = digits(value) + "0"
            elif kind == "unit":
                value = attach_small(payload, prev in self.tens, prev in self.ones)
            elif kind == "unit_sfx":
                n, sfx = payload
                flush(str(attach_small(n, prev in self.tens, prev in self.ones)) + sfx)
            elif kind == "ten":
                value = attach_ten(payload)
            elif kind == "ten_sfx":
                t, sfx = payload
                flush(str(attach_ten(t)) + sfx)
            elif kind == "scale":
                if value is None:
                    value = payload
                elif isinstance(value, str) or value == 0:
                    whole = scaled(value, payload)
                    if whole is None:
                        flush(value)
                        value = payload
                    else:
                        value = whole
                else:
                    value = fold_scale(payload)
            elif kind == "scale_sfx":
                m, sfx = payload
                if value is None:
                    flush(str(m) + sfx)
                elif isinstance(value, str):
                    whole = scaled(value, m)
                    if whole is None:
                        flush(value)
                        flush(str(m) + sfx)
                    else:
                        flush(str(whole) + sfx)
                else:
                    flush(str(fold_scale(m)) + sfx)
            elif kind == "sign":                             # minus / plus: only in front of a number
                flush_pending()
                if nxt_numberish:
                    prefix = payload
                else:
                    flush(cur)
            elif kind == "currency":                         # dollars / cents: only behind a number
                if value is None:
                    flush(cur)
                else:
                    prefix = payload
                    flush(value)
            elif kind == "percent":
                if value is None:
                    flush(cur)
                elif isinstance(payload, dict):              # "per" needs its "cent"
                    if nxt in payload:
                        flush(str(value) + payload[nxt])
                        skip_next = True
                    else:
                        flush(value)
                        flush(cur)
                else:
                    flush(str(value) + payload)
            else:                                            # and / double / triple / point
                if not nxt_numberish:
                    flush_pending()
                    flush(cur)
                elif cur == "and":
                    if prev not in self.multipliers:         # "two hundred and five": the "and" disappears
                        flush_pending()
                        flush(cur)
                elif cur == "point":
                    if nxt in self.decimals or nxt_numeric:
                        value = digits(value) + "."
                elif nxt in self.ones or nxt in self.zeros:  # double / triple + digit word
                    value = digits(value) + str(self.ones.get(nxt, 0)) * (2 if cur == "double" else 3)
                    skip_next = True
                else:
                    flush_pending()
                    flush(cur)
        flush_pending()
        return out

    # ---- string-level passes around the machine ------------------------------------------------------
    def preprocess(self, s: str) -> str:
        # "<number> and a half" -> "<number> point five" (left alone after a non-number)
        pieces = re.split(r"\band\s+a\s+half\b", s)
        rebuilt: List[str] = []
        for i, piece in enumerate(pieces):
            if not piece.strip():
                continue
            rebuilt.append(piece)
            if i + 1 < len(pieces):
                tail = piece.rsplit(maxsplit=2)[-1]
                rebuilt.append("point five" if tail in self.decimals or tail in self.multipliers else "and a half")
        s = " ".join(rebuilt)
        s = re.sub(r"([a-z])([0-9])", r"\1 \2", s)                       # split letters from digits ...
        s = re.sub(r"([0-9])([a-z])", r"\1 \2", s)
        return re.sub(r"([0-9])\s+(st|nd|rd|th|s)\b", r"\1\2", s)       # ... except number suffixes

    def postprocess(self, s: str) -> str:
        # "$2 and ¢7" -> "$2.07"; "$0.07" -> "¢7"; a lone 1 / 1s is written out
        s = re.sub(r"([€£$])([0-9]+) (?:and )?¢([0-9]{1,2})\b",
                   lambda m: f"{m.group(1)}{m.group(2)}.{int(m.group(3)):02d}", s)
        s = re.sub(r"[€£$]0.([0-9]{1,2})\b", lambda m: f"¢{int(m.group(1))}", s)
        return re.sub(r"\b1(s?)\b", r"one\1", s)

    def __call__(self, s: str) -> str:
        s = self.preprocess(s)
        s = " ".join(self.process_words(s.split()))
        return self.postprocess(s)


class EnglishSpellingNormalizer:
    """British -> American spellings, word by word (english.py:443-455).  The table is `english.json`
    beside this file (a data asset: the UK/US spelling list the reference ships; see ASSETS.md), or any
    JSON object passed as `mapping_path`."""

    def __init__(self, mapping_path: Optional[str] = None):
        path = mapping_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "english.json")
        with open(path, encoding="utf-8") as f:
            self.mapping: Dict[str, str] = json.load(f)

    def __call__(self, s: str) -> str:
        return " ".join(self.mapping.get(w, w) for w in s.split())


def _rules(pairs: Iterable[Tuple[str, str]], pattern: str) -> List[Tuple["re.Pattern", str]]:
    return [(re.compile(pattern.format(re.escape(k))), v) for k, v in pairs]


class EnglishTextNormalizer:
    FILLERS = r"\b(hmm|mm|mhm|mmm|uh|um)\b"
    # applied in this order (english.py:459-520)
    WHOLE_WORDS = (("won't", "will not"), ("can't", "can not"), ("let's", "let us"), ("ain't", "aint"),
                   ("y'all", "you all"), ("wanna", "want to"), ("gotta", "got to"), ("gonna", "going to"),
                   ("i'ma", "i am going to"), ("imma", "i am going to"), ("woulda", "would have"),
                   ("coulda", "could have"), ("shoulda", "should have"), ("ma'am", "madam"))
    TITLES = (("mr", "mister"), ("mrs", "missus"), ("st", "saint"), ("dr", "doctor"), ("prof", "professor"),
              ("capt", "captain"), ("gov", "governor"), ("ald", "alderman"), ("gen", "general"), ("sen", "senator"),
              ("rep", "representative"), ("pres", "president"), ("rev", "reverend"), ("hon", "honorable"),
              ("asst", "assistant"), ("assoc", "associate"), ("lt", "lieutenant"), ("col", "colonel"),
              ("jr", "junior"), ("sr", "senior"), ("esq", "esquire"))
    ENDINGS = (("'d been", " had been"), ("'s been", " has been"), ("'d gone", " had gone"), ("'s gone", " has gone"),
               ("'d done", " had done"), ("'s got", " has got"),
               ("n't", " not"), ("'re", " are"), ("'s", " is"), ("'d", " would"), ("'ll", " will"), ("'t", " not"),
               ("'ve", " have"), ("'m", " am"))

    def __init__(self, spelling_path: Optional[str] = None):
        self.ignore_patterns = self.FILLERS
        self.replacers = (_rules(self.WHOLE_WORDS, r"\b{}\b")
                          + _rules(((k, v + " ") for k, v in self.TITLES), r"\b{}\b")
                          + _rules(self.ENDINGS, r"{}\b"))
        self.standardize_numbers = EnglishNumberNormalizer()
        self.standardize_spellings = EnglishSpellingNormalizer(spelling_path)

    def __call__(self, s: str) -> str:
        s = s.lower()
        s = re.sub(r"[<\[][^>\]]*[>\]]", "", s)              # <..> and [..] asides
        s = re.sub(r"\(([^)]+?)\)", "", s)                   # (..) asides
        s = re.sub(self.ignore_patterns, "", s)
        s = re.sub(r"\s+'", "'", s)                          # stray space in front of an apostrophe
        for pattern, replacement in self.replacers:
            s = pattern.sub(replacement, s)
        s = re.sub(r"(\d),(\d)", r"\1\2", s)                 # thousands separators
        s = re.sub(r"\.([^0-9]|$)", r" \1", s)               # full stops that are not decimal points
        s = remove_symbols_and_diacritics(s, keep=".%$¢€£")  # symbols that can belong to a number survive
        s = self.standardize_numbers(s)
        s = self.standardize_spellings(s)
        s = re.sub(r"[.$¢€£]([^0-9])", r" \1", s)            # ... unless no digit follows / precedes them
        s = re.sub(r"([^0-9])%", r"\1 ", s)
        return re.sub(r"\s+", " ", s)
