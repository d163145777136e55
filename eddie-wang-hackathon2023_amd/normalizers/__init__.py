"""Text normalisation applied to hypotheses and references before WER (W/normalizers/__init__.py)."""
from .basic import BasicTextNormalizer
from .english import EnglishNumberNormalizer, EnglishSpellingNormalizer, EnglishTextNormalizer

__all__ = ["BasicTextNormalizer", "EnglishTextNormalizer", "EnglishNumberNormalizer", "EnglishSpellingNormalizer"]
