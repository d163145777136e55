"""CPU tests of the rows either side of the hot path: FLAC decoder, text normalisers
(pinned to the reference's own output, tests/golden/normalizer.json from oracle/gen_golden_text.py), WER,
and the summarize.py host logic."""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from flac_writer import encode_flac  # noqa: E402

FLAC_FIXTURE = "librispeech_1089-134691-0000.flac"      # "HE COULD WAIT NO LONGER", LibriSpeech test-clean (CC BY 4.0)


# ---- FLAC ---------------------------------------------------------------------------------------------
def test_flac_librispeech_clip(golden_dir):
    import whisper_utils as wu
    data = open(os.path.join(golden_dir, FLAC_FIXTURE), "rb").read()
    pcm, rate, bits = wu.decode_flac(data)          # CRC-8/CRC-16 per frame in C, MD5 of the PCM here
    assert (pcm.shape, rate, bits) == ((33360, 1), 16000, 16)
    assert int(pcm.astype(np.int64).sum()) == -95452 and int(np.abs(pcm).max()) == 25185
    assert pcm[1000:1004, 0].tolist() == [52, 40, 53, 43]
    assert hashlib.sha256(pcm.astype("<i2").tobytes()).hexdigest().startswith("8765bbf3602482c6")
    audio = wu.load_audio(os.path.join(golden_dir, FLAC_FIXTURE))
    assert audio.dtype == np.float32 and audio.shape == (33360,)
    np.testing.assert_array_equal(audio, pcm[:, 0].astype(np.float32) / 32768.0)     # what ffmpeg s16le / 32768 gives


def test_flac_corruption_is_detected(golden_dir):
    import native
    import whisper_utils as wu
    data = bytearray(open(os.path.join(golden_dir, FLAC_FIXTURE), "rb").read())
    flipped = bytearray(data)
    flipped[len(data) // 2] ^= 0x10
    with pytest.raises(native.WmError, match="flac"):
        wu.decode_flac(bytes(flipped))
    with pytest.raises(native.WmError, match="flac"):
        wu.decode_flac(bytes(data[: len(data) // 2]))
    with pytest.raises(native.WmError, match="fLaC"):
        wu.decode_flac(b"RIFF" + bytes(data[4:]))
    bad_md5 = bytearray(data)
    bad_md5[4 + 4 + 18] ^= 0xff                      # STREAMINFO signature byte
    with pytest.raises(RuntimeError, match="MD5"):
        wu.decode_flac(bytes(bad_md5))
    pcm, _, _ = wu.decode_flac(bytes(bad_md5), verify_md5=False)
    assert pcm.shape == (33360, 1)


def _smooth(rng, n, nch, bps, wasted=0):
    """Band-limited noise that uses most of the sample range (so LPC / FIXED residuals are small but not tiny)."""
    x = np.cumsum(rng.integers(-300, 301, size=(n, nch)), axis=0)
    x = x - x.mean(axis=0, keepdims=True).astype(np.int64)
    lim = (1 << (bps - 1 - wasted)) - 1
    x = np.clip(x * max(1, lim // 40000), -lim, lim).astype(np.int64)
    return x << wasted


FLAC_CASES = {
    "mono16_fixed_orders": dict(nch=1, bps=16, frames=[dict(block=192, kind=f"fixed{k}") for k in range(5)]),
    "mono16_lpc_rice2_partitions": dict(nch=1, bps=16, frames=[
        dict(block=256, kind="lpc", lpc=dict(coefs=[1900, -900, 20], shift=10, precision=12), rice2=True, porder=3),
        dict(block=512, kind="lpc", lpc=dict(coefs=[7, -3, 1, 2, -1, 1, 0, 1], shift=3, precision=5), porder=4)]),
    "mono16_escape_and_verbatim": dict(nch=1, bps=16, frames=[
        dict(block=256, kind="fixed2", escape=True, porder=2), dict(block=100, kind="verbatim"),
        dict(block=4096, kind="fixed1", rice2=True, escape=True, porder=5)]),
    "mono16_constant_and_wasted": dict(nch=1, bps=16, wasted=3, const_first=True, frames=[
        dict(block=64, kind="constant"), dict(block=576, kind="fixed3", wasted=3), dict(block=33, kind="verbatim", wasted=3)]),
    "stereo16_all_assignments": dict(nch=2, bps=16, frames=[
        dict(block=256, kind="fixed2", assignment=a, porder=1) for a in (1, 8, 9, 10)] + [
        dict(block=1000, kinds=["lpc", "fixed4"], lpc=dict(coefs=[30, -14], shift=4, precision=7), assignment=10)]),
    "stereo24_explicit_fields": dict(nch=2, bps=24, frame0=125, frames=[
        dict(block=300, kind="fixed2", assignment=8, explicit_rate=True),
        dict(block=4096, kind="fixed3", assignment=10, explicit_block=True, bits_from_streaminfo=True),
        dict(block=17, kind="verbatim", assignment=9), dict(block=16, kind="fixed4", assignment=1)] +
        [dict(block=20, kind="fixed1") for _ in range(6)]),
    "mono8_unknown_length_no_md5": dict(nch=1, bps=8, known_total=False, with_md5=False, frames=[
        dict(block=1152, kind="fixed1"), dict(block=200, kind="fixed0", rice2=True)]),
    "five_channels_12bit": dict(nch=5, bps=12, frames=[dict(block=192, kind="fixed2"), dict(block=50, kind="verbatim")]),
}


@pytest.mark.parametrize("name", sorted(FLAC_CASES))
def test_flac_synthetic_streams(name):
    """Streams from tests/flac_writer.py (an independent encoder) that reach every decoder branch."""
    import whisper_utils as wu
    case = FLAC_CASES[name]
    rng = np.random.default_rng(sorted(FLAC_CASES).index(name))
    n = sum(f["block"] for f in case["frames"])
    x = _smooth(rng, n, case["nch"], case["bps"], case.get("wasted", 0))
    if case.get("const_first"):
        x[: case["frames"][0]["block"]] = -1234 << case.get("wasted", 0)
    blob = encode_flac(x, case["bps"], case["frames"], first_frame_number=case.get("frame0", 0),
                       known_total=case.get("known_total", True), with_md5=case.get("with_md5", True))
    pcm, rate, bits = wu.decode_flac(blob)
    assert (rate, bits) == (16000, case["bps"])
    np.testing.assert_array_equal(pcm, x)


# ---- normalisers --------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def normalizer_golden(golden_dir):
    with open(os.path.join(golden_dir, "normalizer.json"), encoding="utf-8") as f:
        return json.load(f)


def test_english_normalizer_matches_reference(normalizer_golden):
    from normalizers import EnglishTextNormalizer
    norm = EnglishTextNormalizer()
    assert len(normalizer_golden["english"]) > 700
    for text, expected in normalizer_golden["english"]:
        assert norm(text) == expected, text


def test_number_normalizer_matches_reference_on_random_word_sequences(normalizer_golden):
    from normalizers import EnglishNumberNormalizer
    norm = EnglishNumberNormalizer()
    assert len(normalizer_golden["numbers"]) == 2500
    for text, expected in normalizer_golden["numbers"]:
        assert norm(text) == expected, text


def test_basic_normalizer_matches_reference(normalizer_golden):
    from normalizers import BasicTextNormalizer
    plain, folded = BasicTextNormalizer(), BasicTextNormalizer(remove_diacritics=True)
    for text, e_plain, e_folded in normalizer_golden["basic"]:
        assert plain(text) == e_plain and folded(text) == e_folded, text
    assert BasicTextNormalizer(split_letters=True)("ab c") == "a b c"


def test_normalizer_examples():
    """The documented behaviours (W/normalizers/english.py:13-21)."""
    from normalizers import EnglishTextNormalizer
    n = EnglishTextNormalizer()
    assert n("$20 million") == "$20000000" and n("twenty dollars") == "$20"
    assert n("one oh one") == "101" and n("the 1960s, the 274th, the 32nd") == "the 1960s the 274th the 32nd"
    assert n("Mr. Brown's colour") == "mister brown is color"
    assert n("two and a half percent") == "2.5%"


# ---- WER ----------------------------------------------------------------------------------------------
def test_wer_known_answers():
    from wer import edit_counts, wer
    assert wer("a b c d", "a b c d") == 0.0
    assert wer("a b c d", "a x c d") == 0.25
    assert wer("a b c d", "a c d") == 0.25
    assert wer("a b c d", "a b b c d") == 0.25
    assert wer("a b c d", "") == 1.0
    assert wer("a b", "x y z") == 1.5                      # insertions can push it above 1
    c = edit_counts("the cat sat on the mat".split(), "the cat sit on mat now".split())
    assert c["substitutions"] + c["deletions"] + c["insertions"] == 3      # two optimal alignments exist
    assert edit_counts("a b c".split(), "a c".split()) == dict(hits=2, substitutions=0, deletions=1, insertions=0)
    assert edit_counts("a c".split(), "a b c".split()) == dict(hits=2, substitutions=0, deletions=0, insertions=1)
    # corpus level: errors and words are pooled (what jiwer.wer does with two lists), not averaged per sentence
    assert wer(["a b c d", "e f"], ["a b c d", "e x"]) == pytest.approx(1 / 6)
    assert wer(["  a   b ", "c"], ["a b", "c"]) == 0.0
    with pytest.raises(ValueError):
        wer(["a", "b"], ["a"])
    with pytest.raises(ValueError):
        wer([""], ["a"])


def test_wer_against_brute_force():
    from wer import edit_counts
    rng = np.random.default_rng(3)

    def brute(r, h):
        from functools import lru_cache

        @lru_cache(None)
        def d(i, j):
            if i == 0 or j == 0:
                return i + j
            return min(d(i - 1, j - 1) + (r[i - 1] != h[j - 1]), d(i - 1, j) + 1, d(i, j - 1) + 1)
        return d(len(r), len(h))
    for _ in range(200):
        r = tuple(rng.integers(0, 4, rng.integers(0, 9)).tolist())
        h = tuple(rng.integers(0, 4, rng.integers(0, 9)).tolist())
        c = edit_counts(r, h)
        assert c["substitutions"] + c["deletions"] + c["insertions"] == brute(r, h)
        assert c["hits"] + c["substitutions"] + c["deletions"] == len(r)
        assert c["hits"] + c["substitutions"] + c["insertions"] == len(h)


# ---- summarize.py host logic ----------------------------------------------------------------------------
def test_summarize_dataset_discovery_and_scoring(golden_dir, tmp_path):
    import summarize as S
    # nested LibriSpeech layout (speaker/chapter) and a flat directory with one transcript file
    src = os.path.join(golden_dir, FLAC_FIXTURE)
    chapter = tmp_path / "nested" / "1089" / "134691"
    chapter.mkdir(parents=True)
    for i in range(3):
        shutil.copy(src, chapter / f"1089-134691-000{i}.flac")
    (chapter / "1089-134691.trans.txt").write_text(
        "1089-134691-0000 HE COULD WAIT NO LONGER\n1089-134691-0001 SECOND LINE\n1089-134691-0002 THIRD LINE HERE\n")
    pairs = S.discover(tmp_path / "nested")
    assert [p.name for p, _ in pairs] == [f"1089-134691-000{i}.flac" for i in range(3)]
    assert [r for _, r in pairs] == ["HE COULD WAIT NO LONGER", "SECOND LINE", "THIRD LINE HERE"]
    flat = tmp_path / "flat"
    flat.mkdir()
    shutil.copy(src, flat / "b-0.flac")
    shutil.copy(src, flat / "a-0.flac")
    (flat / "valid.trans.txt").write_text("a-0 FIRST\nb-0 SECOND\nc-0 NO AUDIO FOR THIS ONE\n")
    assert [(p.name, r) for p, r in S.discover(flat)] == [("a-0.flac", "FIRST"), ("b-0.flac", "SECOND")]
    assert S.clean_hypothesis("Hello, world! Isn't it? Yes.") == "HELLO WORLD ISN'T IT YES"
    assert S.score(["HE COULD WAIT NO LONGER", "MR BROWN PAID TWENTY DOLLARS"],
                   ["he could wait no longer", "mister brown paid $20"]) == 0.0
    assert S.score(["HE COULD WAIT NO LONGER"], ["HE COULD WEIGHT NO LONGER"]) == pytest.approx(0.2)
    args = S.parse_arguments(["--test_trt_llm", "--engine_dir", "e", "--dataset_dir", "d"])
    assert args.test_trt_llm and not args.test_torch and args.data_type == "fp16" and args.checkpoint_file == "./large-v2.pt"


def test_pipelined_evaluation_protocol_with_fakes():
    """summarize.eval_engines_stream / transcribe_dataset_stream: the host-side pipelining protocol on fakes (no GPU, no
    library).  The encoder of batch n + 1 is handed to prefetch() after the language pass of batch n and before its decode
    loop, `loop_ended()` follows every decode loop (round 5: the pass in flight gives back its CU budget from there), collect() is
    called once per prefetch, the first batch is encoded directly, a budget of 0 switches the overlap off, and results come back in
    batch order."""
    import summarize as S
    log = []

    class Enc:
        def __init__(self): self.pending = None
        def get_audio_features_async(self, mel): log.append(("encode", mel)); return ("xa", mel)
        def prefetch(self, mel, budget): assert self.pending is None; log.append(("prefetch", mel, budget)); self.pending = mel
        def collect(self): mel, self.pending = self.pending, None; log.append(("collect", mel)); return ("xa", mel)
        def loop_ended(self): log.append(("loop ended", self.pending))

    class Dec:
        def detect_language(self, xa): log.append(("lang", xa[1])); return ["en"], None
        def main_loop(self, xa): log.append(("loop", xa[1])); return [xa[1]], [0.0], [0.0]
        def post_process(self, tokens, lp, nsp, xa, languages): return [("result", xa[1])]

    out = list(S.eval_engines_stream(Enc(), Dec(), iter(["a", "b", "c"]), cu_budget=24))
    assert out == [[("result", "a")], [("result", "b")], [("result", "c")]]
    assert log == [("encode", "a"), ("lang", "a"), ("prefetch", "b", 24), ("loop", "a"), ("loop ended", "b"),
                   ("collect", "b"), ("lang", "b"), ("prefetch", "c", 24), ("loop", "b"), ("loop ended", "c"),
                   ("collect", "c"), ("lang", "c"), ("loop", "c"), ("loop ended", None)]
    log.clear()
    out = list(S.eval_engines_stream(Enc(), Dec(), iter(["a", "b"]), cu_budget=0))
    assert out == [[("result", "a")], [("result", "b")]]
    assert [e for e in log if e[0] != "loop ended"] == [("encode", "a"), ("lang", "a"), ("loop", "a"), ("encode", "b"), ("lang", "b"), ("loop", "b")]
    assert list(S.eval_engines_stream(Enc(), Dec(), iter([]))) == []


# ---- BPE against an implementation that is not ours ---------------------------------------------------------------------
def test_bpe_encode_matches_independent_implementation_and_known_gpt2_ids(golden_dir):
    """tokenizer.py's byte-pair encoder on the vendored Whisper vocabularies (assets/ASSETS.md) against
    (1) ids produced by HuggingFace `tokenizers` loaded with the same ranks (oracle/gen_golden_bpe.py: an
    independent BPE -- byte-level pre-tokeniser, merges by priority), 440 strings per vocabulary (LibriSpeech
    transcripts in three casings, hand-written punctuation / unicode / whitespace cases, seeded random strings);
    (2) published GPT-2 encodings (the English-only vocabulary IS GPT-2's) written down by hand.
    The reference delegates this to the tiktoken wheel (W/tokenizer.py:125-317)."""
    import json
    import tokenizer as T
    assets = os.path.join(os.path.dirname(os.path.abspath(T.__file__)), "assets")
    with open(os.path.join(golden_dir, "tokenizer_bpe.json"), encoding="utf-8") as f:
        fx = json.load(f)
    for name, multilingual in (("multilingual", True), ("gpt2", False)):
        tk = T.Tokenizer.from_vocab(os.path.join(assets, name + ".tiktoken"), multilingual)
        assert tk.n_base == (50257 if multilingual else 50256)
        bad = [(text, tk.encode(text), ids) for text, ids in fx[name] if tk.encode(text) != ids]
        assert not bad, bad[:3]
        for text, ids in fx[name][:60]:
            assert tk.decode(ids) == text
    gp = T.Tokenizer.from_vocab(os.path.join(assets, "gpt2.tiktoken"), False)
    for text, ids in fx["gpt2_known"]:
        assert gp.encode(text) == ids, text
    assert sum(len(ids) for _, ids in fx["multilingual"]) > 10000          # not a vacuous fixture


def test_clips_are_batched_by_duration_and_dealt_over_the_ranks(golden_dir, tmp_path):
    """summarize.plan_batches (length-sorted chunks): durations come from the headers (FLAC STREAMINFO,
    .npy shape, wav header) without decoding; clips of similar length share a batch; the batches are dealt
    round-robin over the ranks so that every rank sees short and long batches; nothing is lost or duplicated."""
    import wave
    import summarize as S
    rng = np.random.Generator(np.random.Philox(3))
    want = {}
    for k, n in enumerate([16000, 4000, 52000, 9000, 30000, 2500, 41000]):
        x = _smooth(rng, n, 1, 16)
        (tmp_path / f"a{k}.flac").write_bytes(encode_flac(x, 16, [dict(block=min(4096, n - o), kind="fixed2") for o in range(0, n, 4096)]))
        want[f"a{k}.flac"] = n / 16000.0
    np.save(tmp_path / "b0.npy", np.zeros(24000, dtype=np.float32)); want["b0.npy"] = 1.5
    with wave.open(str(tmp_path / "c0.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(bytes(2 * 12000))
    want["c0.wav"] = 0.75
    shutil.copy(os.path.join(golden_dir, FLAC_FIXTURE), tmp_path / "d0.flac"); want["d0.flac"] = 33360 / 16000.0
    for name, sec in want.items():
        assert abs(S.clip_seconds(tmp_path / name) - sec) < 1e-9, name
    pairs = [(tmp_path / name, name.upper()) for name in sorted(want)]
    single = S.plan_batches(pairs, 4)
    flat = [p for b in single for p in b]
    assert sorted(map(str, (p[0] for p in flat))) == sorted(map(str, (p[0] for p in pairs)))       # a permutation
    secs = [want[p[0].name] for p in flat]
    assert secs == sorted(secs) and [len(b) for b in single] == [4, 4, 2]
    assert all(p[1] == p[0].name.upper() for p in flat)                                             # references stay attached
    per_rank = [S.plan_batches(pairs, 2, r, 2) for r in range(2)]
    assert [len(b) for b in per_rank[0]] == [2, 2, 2] and [len(b) for b in per_rank[1]] == [2, 2]
    seen = sorted(str(p[0]) for r in per_rank for b in r for p in b)
    assert seen == sorted(str(p[0]) for p in pairs)
    # every rank gets short AND long batches (a contiguous slice of the sorted list would not)
    means = [[np.mean([want[p[0].name] for p in b]) for b in r] for r in per_rank]
    assert min(means[0]) < 0.6 and max(means[0]) > 2.0 and min(means[1]) < 1.0 and max(means[1]) > 1.8


def test_clip_without_a_duration_in_its_header_sorts_among_the_others(tmp_path):
    """clip_seconds' fallback used to return BYTES on the seconds scale, so one odd header sent its clip behind
    every 30 s clip.  A FLAC whose STREAMINFO says total_samples = 0 (allowed: "unknown") now gets size / 16 kB per second."""
    import summarize as S
    rng = np.random.Generator(np.random.Philox(5))
    x = _smooth(rng, 48000, 1, 16)                                   # 3 s
    blob = bytearray(encode_flac(x, 16, [dict(block=4096, kind="fixed2") for _ in range(0, 48000 - 4095, 4096)] + [dict(block=48000 % 4096, kind="fixed2")] if 48000 % 4096 else
                                 [dict(block=4096, kind="fixed2") for _ in range(0, 48000, 4096)]))
    assert abs(S.clip_seconds(_write(tmp_path, "known.flac", bytes(blob))) - 3.0) < 1e-9
    blob[8 + 13] &= 0xF0                                            # total_samples (36 bits) := 0
    blob[8 + 14:8 + 18] = bytes(4)
    est = S.clip_seconds(_write(tmp_path, "unknown.flac", bytes(blob)))
    assert 0.5 < est < 12.0                                          # seconds, not bytes (the file is ~ 50-100 kB)


def _write(tmp_path, name, data):
    path = tmp_path / name
    path.write_bytes(data)
    return path
