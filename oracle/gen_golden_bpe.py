"""Generate tests/golden/tokenizer_bpe.json: BPE encodings from an implementation that is NOT ours.

Run once in the build container:   python oracle/gen_golden_bpe.py
Test infrastructure, not product code.  The reference delegates BPE to the `tiktoken` wheel (W/tokenizer.py:125-317,
W/decoding.py:423-456), which is not installable here; eddie-wang-hackathon2023_amd/tokenizer.py re-implements the
published algorithm.  To pin that implementation to something it did not produce itself, this script builds the
same vocabulary inside HuggingFace `tokenizers` (an independent BPE in Rust: byte-level pre-tokeniser with GPT-2's
split pattern, merges applied in priority order) and records ITS ids for a few hundred strings.  The merge list is
recovered from the rank file in the standard way: a token of rank r splits into the two parts that byte-pair merging
with ranks < r leaves of it.  The fixture holds strings and ids only.
"""
import base64
import json
import os
import random

from tokenizers import Tokenizer, decoders, models, pre_tokenizers

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W = "/root/reference/tensorrt_llm_july-release-v1/examples/whisper"
OUT = os.path.join(ROOT, "tests", "golden", "tokenizer_bpe.json")


def bytes_to_unicode():
    """GPT-2's printable stand-ins for the 256 byte values (published with GPT-2's encoder.py)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


def split_by_lower_ranks(ranks, token, max_rank):
    parts = [bytes([b]) for b in token]
    while True:
        best, at = None, -1
        for i in range(len(parts) - 1):
            r = ranks.get(parts[i] + parts[i + 1])
            if r is not None and r < max_rank and (best is None or r < best):
                best, at = r, i
        if best is None:
            return parts
        parts[at:at + 2] = [parts[at] + parts[at + 1]]


def hf_tokenizer(path):
    ranks = {}
    with open(path) as f:
        for line in f:
            if line.strip():
                tok, rank = line.split()
                ranks[base64.b64decode(tok)] = int(rank)
    b2u = bytes_to_unicode()
    uni = lambda bs: "".join(b2u[b] for b in bs)
    vocab = {uni(t): r for t, r in ranks.items() if t}
    merges = []
    for t, r in sorted(ranks.items(), key=lambda kv: kv[1]):
        if len(t) <= 1:        # single bytes; and the multilingual file's one empty entry ("= 50256", a placeholder rank)
            continue
        parts = split_by_lower_ranks(ranks, t, r)
        assert len(parts) == 2, (t, parts)
        merges.append((uni(parts[0]), uni(parts[1])))
    tk = Tokenizer(models.BPE(vocab=vocab, merges=merges))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)
    tk.decoder = decoders.ByteLevel()
    return tk


HAND = [
    "Hello world", "Hello, world!", " Hello world, it's 42 degrees! ♪ 你好 <3", "", " ", "  ", "   leading and trailing   ",
    "He could wait no longer.", "I won't, can't, y'all shouldn't've; he's gone, she'd been, they're, we'll, I'm",
    "naïve café Straße Ærøskøbing", "Привет, мир! Как дела?", "こんにちは世界、元気ですか？", "안녕하세요 세계", "مرحبا بالعالم", "שלום עולם",
    "emoji 🙂🙃 family 👨‍👩‍👧‍👦 flags 🇩🇪🇯🇵", "tabs\tand\nnewlines\r\n\n  mixed   spaces", "1234567890 3.14159 1,000,000 0x1F 1e-9",
    "snake_case CamelCase kebab-case SCREAMING", "$20 million, £3.50 or €0.99, 5% off!!!", "a" * 40, "ab" * 33, " the" * 12,
    "(laughs) [noise] <unk> {braces} |pipes| \\backslash/ ~tilde~ `tick`", "...---... ???!!! ;;;:::", "ɑ̃ ẽ ĩ õ ũ combining é vs é",
    "\u00a0non-breaking\u2003em-space\u200bzero-width", "ＦＵＬＬＷＩＤＴＨ ｔｅｘｔ", "Ω≈ç√∫˜µ≤≥÷ ¡™£¢∞§¶•ªº",
]


def cases(seed):
    with open(os.path.join(W, "LibriSpeech", "valid-clean", "valid.trans.txt")) as f:
        libri = [line.split(" ", 1)[1].rstrip("\n") for line in f if " " in line]
    out = list(HAND)
    for s in libri:
        out += [s, " " + s.capitalize() + ".", s.lower()]
    rng = random.Random(seed)
    alphabet = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 .,'-!?\n\t" + "éüßñçøåœæ“”‘’—…" + "中文日本語한국어"
    for _ in range(150):
        out.append("".join(rng.choice(alphabet) for _ in range(rng.randint(1, 60))))
    return out


if __name__ == "__main__":
    fixture = {}
    for name in ("multilingual", "gpt2"):
        tk = hf_tokenizer(os.path.join(W, "assets", name + ".tiktoken"))
        rows = []
        for text in cases(5):
            enc = tk.encode(text, add_special_tokens=False)
            assert tk.decode(enc.ids, skip_special_tokens=False) == text or "\ufffd" in tk.decode(enc.ids), text
            rows.append([text, enc.ids])
        fixture[name] = rows
    # published GPT-2 encodings (the gpt2 vocabulary of the English-only models IS GPT-2's): known answers by hand
    fixture["gpt2_known"] = [["Hello world", [15496, 995]], ["Hello, world!", [15496, 11, 995, 0]], [" ", [220]],
                             ["<|endoftext|>"[2:-2], [437, 1659, 5239]]]
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(fixture, f, ensure_ascii=False, indent=0)
    print(OUT, {k: len(v) for k, v in fixture.items()}, os.path.getsize(OUT), "bytes")
