"""Language-independent transcript clean-up (behaviour of W/normalizers/basic.py:1-76).

Everything is done per Unicode code point after a compatibility decomposition: marks / symbols /
punctuation (general categories M*, S*, P*) become spaces; with `remove_diacritics` the combining
marks (Mn) vanish instead and a handful of letters that NFKD leaves alone are spelled out.
"""
import re
import unicodedata

# letters NFKD does not split into base + mark (W/normalizers/basic.py:7-24)
_UNSPLIT_LETTERS = dict(zip("œŒøØæÆßẞđĐðÐþÞłŁ",
                            ("oe", "OE", "o", "O", "ae", "AE", "ss", "SS", "d", "D", "d", "D", "th", "th", "l", "L")))
_BRACKETED = re.compile(r"[<\[][^>\]]*[>\]]")
_PARENTHESISED = re.compile(r"\(([^)]+?)\)")
_BLANKS = re.compile(r"\s+")


def _is_mark_symbol_punct(ch: str) -> bool:
    return unicodedata.category(ch)[0] in "MSP"


def remove_symbols_and_diacritics(s: str, keep: str = "") -> str:
    """NFKD, then per character: kept verbatim if in `keep`; spelled out if NFKD cannot split it; dropped
    if it is a non-spacing mark; a space if any other mark / symbol / punctuation (basic.py:27-44)."""
    out = []
    for ch in unicodedata.normalize("NFKD", s):
        if ch in keep:
            out.append(ch)
        elif ch in _UNSPLIT_LETTERS:
            out.append(_UNSPLIT_LETTERS[ch])
        elif unicodedata.category(ch) == "Mn":
            continue
        elif _is_mark_symbol_punct(ch):
            out.append(" ")
        else:
            out.append(ch)
    return "".join(out)


def remove_symbols(s: str) -> str:
    """NFKC, marks / symbols / punctuation -> space, diacritics stay (basic.py:47-54)."""
    return "".join(" " if _is_mark_symbol_punct(ch) else ch for ch in unicodedata.normalize("NFKC", s))


class BasicTextNormalizer:
    def __init__(self, remove_diacritics: bool = False, split_letters: bool = False):
        self.clean = remove_symbols_and_diacritics if remove_diacritics else remove_symbols
        self.split_letters = split_letters

    def __call__(self, s: str) -> str:
        s = _PARENTHESISED.sub("", _BRACKETED.sub("", s.lower()))     # drop <..>, [..] and (..) asides
        s = self.clean(s).lower()
        if self.split_letters:
            import regex                                              # grapheme clusters (\X)
            s = " ".join(regex.findall(r"\X", s, regex.U))
        return _BLANKS.sub(" ", s)
