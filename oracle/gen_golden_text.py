"""Generate tests/golden/normalizer.json by running THE REFERENCE's text normalisers in the build container.

Run once, here, where /root/reference exists:   python oracle/gen_golden_text.py
Imports W/normalizers (basic.py, english.py + english.json) for real; `more_itertools` is not in this image,
so its one used helper (`windowed(seq, 3)`, a sliding window) is provided by a stand-in module.  The fixture
holds input strings and the reference's outputs only.  Test infrastructure, not product code.

Inputs: the 87 LibriSpeech valid-clean transcripts the reference carries (W/LibriSpeech/valid-clean/
valid.trans.txt, public-domain text), hand-written number / currency / contraction cases, and seeded random
word sequences over the number vocabulary (the accumulator machine's state space).
"""
import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W = "/root/reference/tensorrt_llm_july-release-v1/examples/whisper"
OUT = os.path.join(ROOT, "tests", "golden", "normalizer.json")

mi = types.ModuleType("more_itertools")
mi.windowed = lambda seq, n: (tuple(seq[i:i + n]) for i in range(len(seq) - n + 1))
sys.modules["more_itertools"] = mi
sys.path.insert(0, W)
from normalizers import BasicTextNormalizer, EnglishTextNormalizer  # noqa: E402  (the reference's package)
from normalizers.english import EnglishNumberNormalizer  # noqa: E402

HAND = [
    "He paid $20 million, or twenty million dollars, in 1960s money.", "one oh one", "two hundred and five",
    "nineteen eighty four", "twenty twenty one was the 2nd year; the twenty-first century", "three and a half million",
    "two and a half", "half and a half", "minus five degrees, plus six", "positive thinking", "negative two point five",
    "five percent and six per cent, per se", "double oh seven and triple nine", "point five", "what's the point",
    "the first, the second and the forty-third", "nine hundred ninety-nine thousand nine hundred ninety nine",
    "a hundred thousandths", "one's own ones", "$2 and 7 cents", "two dollars and seven cents", "£3.50 or €0.99",
    "zero point zero five dollars", "seven cents", "I won't, can't, y'all shouldn't've; he's gone, she'd been",
    "Mr. Smith, Mrs. Jones and Dr. Who met St. Peter Jr. Esq.", "uh, um... hmm, (laughs) [noise] <unk> okay",
    "colour centre theatre organise travelled", "Ærøskøbing naïve café Straße", "1,000,000 and 3.14159 and 2.0",
    "ten sixes are sixty", "twelfth night, the fifties, the 50s", "it's 10 o'clock at 192.168.1.1",
    "one hundred and one dalmatians", "a million and one", "two million three hundred thousand and four",
    "eleven hundred", "one two three four", "thirty first", "hundredth", "zero", "oh", "o", "and", "double", "point",
    "twenty first century fox", "five billions", "the nineties", "1st 2nd 3rd 4th 5 th 6 s", "3rd2",
    "point one four", "one point", "one point hundred", "two hundred point five", "minus", "minus and plus",
    "six pounds ten", "per cent", "ten per", "ten per cent", "ten percent", "percent", "dollars", "twenty one dollars fifty cents",
    "", " ", "and a half", "one and a half and a half", "thousand and a half",
]


def fuzz_cases(n: int, seed: int):
    nn = EnglishNumberNormalizer()
    vocab = sorted(nn.words) + ["the", "a", "half", "cat", "7", "12", "3.5", "$5", "-2", "100", "0", "1", "+4", "¢50"]
    rng = random.Random(seed)
    for _ in range(n):
        yield " ".join(rng.choice(vocab) for _ in range(rng.randint(1, 7)))


if __name__ == "__main__":
    eng, basic, basic_d = EnglishTextNormalizer(), BasicTextNormalizer(), BasicTextNormalizer(remove_diacritics=True)
    nn = EnglishNumberNormalizer()
    with open(os.path.join(W, "LibriSpeech", "valid-clean", "valid.trans.txt")) as f:
        libri = [line.split(" ", 1)[1].rstrip("\n") for line in f if " " in line]
    text_inputs = libri + HAND
    fixture = {
        "english": [[s, eng(s)] for s in text_inputs + list(fuzz_cases(600, 11))],
        "basic": [[s, basic(s), basic_d(s)] for s in HAND + libri[:10]],
        "numbers": [[s, nn(s)] for s in list(fuzz_cases(2500, 7))],
    }
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(fixture, f, ensure_ascii=False, indent=0)
    print(OUT, {k: len(v) for k, v in fixture.items()}, os.path.getsize(OUT), "bytes")
