"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF in the build container.

Run once, here, where /root/reference exists:   python oracle/gen_golden.py
The outputs (small .npz: seeds, shapes and expected outputs only) are committed; the reference's
Python can not and does not travel to the GPU box.  Test infrastructure, not product code.

What is imported from the reference (W/ = /root/reference/tensorrt_llm_july-release-v1/examples/whisper):
  * W/torch_model.py  (Whisper, ModelDimensions, install_kv_cache_hooks)  -- real import.
  * W/decoding.py     (ApplyTimestampRules, SuppressBlank, SuppressTokens, GreedyDecoder, MaximumLikelihoodRanker,
    WhisperDecoding.main_loop / post_process as unbound functions) --
    imported with permissive stub modules for tiktoken / tensorrt / tensorrt_llm / build /
    tokenizer, none of which the logit rules touch.
  * W/assets/*.tiktoken  -- read (not copied) to pin our BPE + suppress list.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W = "/root/reference/tensorrt_llm_july-release-v1/examples/whisper"
OUT = os.environ.get("WM_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")      # WM_GOLDEN_OUT: a scratch tree, to check that the fixtures regenerate
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "eddie-wang-hackathon2023_amd"))

from oracle.whisper_oracle import (Dims, MICRO, synthetic_state_dict, synthetic_mel)  # noqa: E402
from oracle import decoding_rules as DR  # noqa: E402


def import_reference_model():
    sys.path.insert(0, W)
    import torch_model  # the reference's file
    return torch_model


def import_reference_decoding():
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (), {})
    for name in ["tiktoken", "tensorrt", "tensorrt_llm", "tensorrt_llm.runtime",
                 "tensorrt_llm.runtime.session", "tensorrt_llm.logger", "tensorrt_llm._utils",
                 "build", "tokenizer"]:
        sys.modules[name] = _Any(name)
    sys.path.insert(0, W)
    import decoding  # the reference's file
    for name in ["build", "tokenizer"]:
        del sys.modules[name]
    return decoding


def run_reference_model(tm, dims: Dims, sd, mel, prompt, n_steps, half_input: bool):
    model = tm.Whisper(tm.ModelDimensions(**dims.to_dict()))
    missing = model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    model.eval()
    x = mel.half() if half_input else mel.float()
    out = {}
    with torch.no_grad():
        xa = model.encoder(x)
        out["xa"] = xa.float().numpy()
        cache, hooks = model.install_kv_cache_hooks()
        tokens = torch.tensor([prompt] * mel.shape[0])
        cur = tokens
        logits_all, ids, margins = [], [], []
        for _ in range(n_steps):
            logits = model.decoder(cur, xa, kv_cache=cache)
            last = logits[:, -1].float()
            t2 = last.topk(2, dim=-1).values
            margins.append((t2[:, 0] - t2[:, 1]).numpy())
            nxt = last.argmax(-1)
            logits_all.append(logits.float().numpy())
            ids.append(nxt.numpy())
            cur = nxt[:, None]
        # cross K/V as the hooks stored them (key/value Linear outputs on xa), layer 0 and last
        blk0, blkL = model.decoder.blocks[0], model.decoder.blocks[-1]
        out["cross_k0"] = cache[blk0.cross_attn.key].float().numpy()
        out["cross_v0"] = cache[blk0.cross_attn.value].float().numpy()
        out["cross_vL"] = cache[blkL.cross_attn.value].float().numpy()
        out["self_k0"] = cache[blk0.attn.key].float().numpy()
        out["self_vL"] = cache[blkL.attn.value].float().numpy()
        for h in hooks:
            h.remove()
    out["prefill_logits"] = logits_all[0]
    out["step_logits"] = np.stack([l[:, 0] for l in logits_all[1:]], axis=1)
    out["ids"] = np.stack(ids, axis=1)
    out["margins"] = np.stack(margins, axis=1)
    return out


def gen_model_fixture(tm):
    dims, seed, mel_seed = MICRO, 7, 1234
    prompt, n_steps, B = [5, 17, 900], 6, 2
    sd = synthetic_state_dict(dims, seed)
    mel = synthetic_mel(B, 2 * dims.n_audio_ctx, dims.n_mels, mel_seed)
    fix = {"dims": np.array(list(dims.to_dict().values()), dtype=np.int64),
           "dims_keys": np.array(list(dims.to_dict().keys())),
           "seed": seed, "mel_seed": mel_seed, "batch": B, "prompt": np.array(prompt), "n_steps": n_steps}
    for tag, half in (("f32", False), ("f16", True)):
        r = run_reference_model(tm, dims, sd, mel, prompt, n_steps, half)
        for k, v in r.items():
            fix[f"{tag}_{k}"] = v.astype(np.float32) if v.dtype.kind == "f" else v
    np.savez_compressed(os.path.join(OUT, "model_micro.npz"), **fix)
    print("model_micro.npz: ids f32", fix["f32_ids"].tolist(), "f16", fix["f16_ids"].tolist(),
          "min margin", float(fix["f32_margins"].min()))


def gen_tiny_en_shape_fixture(tm):
    """One fixture at a REAL Whisper shape (SURVEY 8c "one tiny.en-shaped case"): 384 wide, 6 heads, 1500 audio
    positions (mel 3000 frames), gpt2 vocabulary of 51 864, 448 text positions -- depth cut to 2 + 2 layers so that
    the reference's CPU run and the oracle's stay in seconds.  Kept small: every 10th row of the encoder output and of
    layer 0's cross K / V, the top-64 logits (values + ids) per step, greedy ids and margins."""
    dims = Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2)
    seed, mel_seed, B, n_steps = 33, 777, 1, 5
    prompt = [50257, 50362, 1169]                   # <|startoftranscript|>, <|notimestamps|> of the gpt2 vocabulary, a text token
    sd = synthetic_state_dict(dims, seed)
    mel = synthetic_mel(B, 2 * dims.n_audio_ctx, dims.n_mels, mel_seed)
    rows = np.arange(0, dims.n_audio_ctx, 10)
    fix = {"dims": np.array(list(dims.to_dict().values()), dtype=np.int64),
           "dims_keys": np.array(list(dims.to_dict().keys())),
           "seed": seed, "mel_seed": mel_seed, "batch": B, "prompt": np.array(prompt), "n_steps": n_steps, "rows": rows}
    for tag, half in (("f32", False), ("f16", True)):
        r = run_reference_model(tm, dims, sd, mel, prompt, n_steps, half)
        store = np.float32 if tag == "f32" else np.float16      # fp16-mode outputs ARE fp16 values
        fix[f"{tag}_xa"] = r["xa"][:, rows].astype(store)
        fix[f"{tag}_cross_k0"] = r["cross_k0"][:, rows].astype(store)
        fix[f"{tag}_cross_v0"] = r["cross_v0"][:, rows].astype(store)
        fix[f"{tag}_cross_vL"] = r["cross_vL"][:, rows].astype(store)
        last = np.concatenate([r["prefill_logits"][:, -1:], r["step_logits"]], axis=1)        # [B, n_steps, V]
        top = np.argsort(-last, axis=-1, kind="stable")[..., :64]
        fix[f"{tag}_top_ids"] = top.astype(np.int32)
        fix[f"{tag}_top_logits"] = np.take_along_axis(last, top, axis=-1).astype(np.float32)
        fix[f"{tag}_logit_checksum"] = np.abs(last).sum(axis=-1, dtype=np.float64)
        fix[f"{tag}_ids"], fix[f"{tag}_margins"] = r["ids"], r["margins"].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "model_tiny_en_shape.npz"), **fix)
    print("model_tiny_en_shape.npz: ids f32", fix["f32_ids"].tolist(), "f16", fix["f16_ids"].tolist(),
          "min margin", float(fix["f32_margins"].min()), os.path.getsize(os.path.join(OUT, "model_tiny_en_shape.npz")), "bytes")


def gen_op_fixtures(tm):
    """Per-op pins: attention core (the identity-weights trick of R/tests/test_layer.py:616-625 is
    unnecessary here because qkv_attention is callable on its own), LayerNorm, conv1d, GELU."""
    rng = np.random.Generator(np.random.Philox(11))
    fix = {}
    mha = tm.MultiHeadAttention(128, 2)
    q = torch.from_numpy(rng.standard_normal((2, 5, 128)).astype(np.float32))
    k = torch.from_numpy(rng.standard_normal((2, 9, 128)).astype(np.float32))
    v = torch.from_numpy(rng.standard_normal((2, 9, 128)).astype(np.float32))
    mask = torch.full((9, 9), -np.inf).triu_(1)
    with torch.no_grad():
        o, _ = mha.qkv_attention(q, k, v, None)
        fix.update(attn_q=q.numpy(), attn_k=k.numpy(), attn_v=v.numpy(), attn_out=o.numpy())
        oc, _ = mha.qkv_attention(k, k, v, mask)
        fix.update(attn_causal_out=oc.numpy())
        lnm = tm.LayerNorm(128)
        lnm.weight.data = torch.from_numpy(rng.uniform(0.5, 1.5, 128).astype(np.float32))
        lnm.bias.data = torch.from_numpy(rng.uniform(-0.5, 0.5, 128).astype(np.float32))
        x = torch.from_numpy((rng.standard_normal((4, 128)) * 3 + 1).astype(np.float32))
        fix.update(ln_x=x.numpy(), ln_w=lnm.weight.numpy(), ln_b=lnm.bias.numpy(),
                   ln_out=lnm(x).numpy(), ln_out_f16=lnm(x.half()).float().numpy())
        c1 = tm.Conv1d(8, 16, kernel_size=3, padding=1)
        c2 = tm.Conv1d(16, 16, kernel_size=3, stride=2, padding=1)
        # the layers' parameters from THIS function's generator (torch's default init draws from the global generator, whose state
        # here depends on everything constructed before: the fixture would not regenerate), in the default init's range 1 / sqrt(fan_in)
        for c in (c1, c2):
            bound = 1.0 / np.sqrt(c.in_channels * c.kernel_size[0])
            c.weight.data = torch.from_numpy(rng.uniform(-bound, bound, tuple(c.weight.shape)).astype(np.float32))
            c.bias.data = torch.from_numpy(rng.uniform(-bound, bound, tuple(c.bias.shape)).astype(np.float32))
        xin = torch.from_numpy(rng.standard_normal((2, 8, 20)).astype(np.float32))
        y1 = torch.nn.functional.gelu(c1(xin))
        y2 = torch.nn.functional.gelu(c2(y1))
        fix.update(conv_x=xin.numpy(), conv1_w=c1.weight.numpy(), conv1_b=c1.bias.numpy(),
                   conv2_w=c2.weight.numpy(), conv2_b=c2.bias.numpy(), conv1_out=y1.numpy(),
                   conv2_out=y2.numpy())
        g = torch.linspace(-6, 6, 97)
        fix.update(gelu_x=g.numpy(), gelu_out=torch.nn.functional.gelu(g).numpy())
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **fix)
    print("ops.npz written")


class _Tok:
    """What the reference's filters read from a tokenizer (decoding.py:146-199, 209)."""
    def __init__(self, ids, blank):
        self.no_timestamps, self.timestamp_begin, self.eot = ids.no_timestamps, ids.timestamp_begin, ids.eot
        self._blank = list(blank)

    def encode(self, text):
        assert text == " "
        return list(self._blank)


def gen_rules_fixture(dec):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "wm_tokenizer", os.path.join(ROOT, "eddie-wang-hackathon2023_amd", "tokenizer.py"))
    OurTok = importlib.util.module_from_spec(spec)
    sys.modules["wm_tokenizer"] = OurTok
    spec.loader.exec_module(OurTok)
    tk = OurTok.Tokenizer.from_vocab(os.path.join(W, "assets", "multilingual.tiktoken"), True, "en", "transcribe")
    ids = DR.MULTILINGUAL
    assert ids.n_vocab == tk.n_vocab == 51865 and ids.timestamp_begin == tk.timestamp_begin
    # suppress list exactly as WhisperDecoding._get_suppress_tokens builds it (decoding.py:394-421)
    sup = sorted(set(list(tk.non_speech_tokens) + [tk.transcribe, tk.translate, tk.sot, tk.sot_prev,
                                                  tk.sot_lm, tk.no_speech]))
    blank = list(tk.blank_tokens()) + [tk.eot]
    ftok = _Tok(ids, tk.blank_tokens())
    filters = [dec.SuppressBlank(ftok, 3), dec.SuppressTokens(sup), dec.ApplyTimestampRules(ftok, 3, 50)]
    greedy = dec.GreedyDecoder(0.0, ids.eot)
    tb, V = ids.timestamp_begin, ids.n_vocab
    cases = DR.golden_rule_cases(ids)
    fix = {"suppress": np.array(sup), "blank": np.array(blank), "n_cases": len(cases)}
    for c, (toks, logits) in enumerate(cases):
        lt = torch.from_numpy(logits.copy())[None]
        tt = torch.from_numpy(toks)[None]
        for f in filters:
            f.apply(lt, tt)
        s = torch.zeros(1)
        new_tokens, done = greedy.update(tt, lt.clone(), s)
        fix[f"c{c}_tokens"] = toks
        fix[f"c{c}_logits_checksum"] = float(np.abs(logits).sum(dtype=np.float64))
        fix[f"c{c}_filtered_isinf"] = np.packbits(torch.isinf(lt[0]).numpy())
        fix[f"c{c}_next"] = int(new_tokens[0, -1])
        fix[f"c{c}_sumlp"] = float(s[0])
        fix[f"c{c}_done"] = bool(done)
    np.savez_compressed(os.path.join(OUT, "decoding_rules.npz"), **fix)
    print("decoding_rules.npz:", len(cases), "cases; next tokens",
          [fix[f"c{c}_next"] for c in range(len(cases))][:12])

    # tokenizer pins
    gp = OurTok.Tokenizer.from_vocab(os.path.join(W, "assets", "gpt2.tiktoken"), False)
    text = " Hello world, it's 42 degrees! ♪ 你好 <3"
    np.savez_compressed(os.path.join(OUT, "tokenizer.npz"),
                        multilingual_non_speech=np.array(tk.non_speech_tokens),
                        gpt2_non_speech=np.array(gp.non_speech_tokens),
                        multilingual_blank=np.array(tk.blank_tokens()),
                        gpt2_blank=np.array(gp.blank_tokens()),
                        multilingual_sot_sequence=np.array(tk.sot_sequence),
                        multilingual_specials=np.array([tk.eot, tk.sot, tk.translate, tk.transcribe,
                                                        tk.sot_lm, tk.sot_prev, tk.no_speech,
                                                        tk.no_timestamps, tk.timestamp_begin, tk.n_vocab]),
                        sample_text=np.array(text), sample_ids=np.array(tk.encode(text)))
    print("tokenizer.npz written")


def gen_rules_nots_fixture(dec):
    """DecodingOptions.without_timestamps (round 4): the reference then builds SuppressBlank and SuppressTokens but NO
    ApplyTimestampRules (W/decoding.py:332-348), and <|notimestamps|> joins the start sequence (sample_begin = 4).  The cases of
    DR.golden_rule_cases with that token inserted, through the reference's own two filter classes and GreedyDecoder.update."""
    tk_ids = DR.MULTILINGUAL
    base = np.load(os.path.join(OUT, "decoding_rules.npz"))
    sup, blank = base["suppress"].tolist(), base["blank"].tolist()
    ftok = _Tok(tk_ids, blank[:-1])
    filters = [dec.SuppressBlank(ftok, 4), dec.SuppressTokens(sup)]
    greedy = dec.GreedyDecoder(0.0, tk_ids.eot)
    fix = {}
    cases = DR.golden_rule_cases(tk_ids)
    for c, (toks, logits) in enumerate(cases):
        toks = np.concatenate([toks[:3], [tk_ids.no_timestamps], toks[3:]])
        lt = torch.from_numpy(logits.copy())[None]
        tt = torch.from_numpy(toks)[None]
        for f in filters:
            f.apply(lt, tt)
        s = torch.zeros(1)
        new_tokens, done = greedy.update(tt, lt.clone(), s)
        fix[f"c{c}_next"] = int(new_tokens[0, -1])
        fix[f"c{c}_sumlp"] = float(s[0])
        fix[f"c{c}_done"] = bool(done)
    fix["n_cases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "decoding_rules_nots.npz"), **fix)
    print("decoding_rules_nots.npz:", len(cases), "cases; next tokens", [fix[f"c{c}_next"] for c in range(len(cases))][:12])


def gen_sampling_fixture(dec):
    """The sampling path of the decode loop, produced by the reference's own classes:
      (1) GreedyDecoder(temperature=0.7).update (W/decoding.py:274-300) on seeded CPU logits under torch.manual_seed,
          three consecutive steps with EOT stickiness;
      (2) MaximumLikelihoodRanker(length_penalty in {None, 0.6}).rank (:92-115) on fixed candidate groups;
      (3) WhisperDecoding.main_loop (:785-821) + post_process (:827-878) as unbound functions on a stand-in `self`
          carrying best_of = 3, temperature 0.7, the reference's filters / decoder / ranker, and a `decode` that returns
          DR.sampling_logits(step) -- two utterances x three candidates, until every row has sampled EOT.
    Only expected OUTPUTS are stored; the inputs are regenerated from seeds (DR.sampling_logits)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "wm_tokenizer", os.path.join(ROOT, "eddie-wang-hackathon2023_amd", "tokenizer.py"))
    OurTok = sys.modules.get("wm_tokenizer")
    if OurTok is None:
        OurTok = importlib.util.module_from_spec(spec)
        sys.modules["wm_tokenizer"] = OurTok
        spec.loader.exec_module(OurTok)
    tk = OurTok.Tokenizer.from_vocab(os.path.join(W, "assets", "multilingual.tiktoken"), True, "en", "transcribe")
    ids = DR.MULTILINGUAL
    V = ids.n_vocab
    fix = {}

    # (1) one decoder, three updates in a row (the generator state carries over, as in a loop)
    greedy = dec.GreedyDecoder(0.7, ids.eot)
    torch.manual_seed(1234)
    tokens = torch.tensor([[ids.sot, ids.lang0, ids.transcribe]] * 5)
    tokens[3, -1] = ids.eot                       # a row that is already finished: stays at EOT, adds no log-prob
    sum_lp = torch.zeros(5)
    for step in range(3):
        lg = torch.from_numpy(DR.sampling_logits(step + 10, 5, 1)[:, 0].copy())
        tokens, done = greedy.update(tokens, lg, sum_lp)
        fix[f"upd{step}_next"] = tokens[:, -1].numpy().copy()
        fix[f"upd{step}_sumlp"] = sum_lp.numpy().copy()
        fix[f"upd{step}_done"] = bool(done)
    ftok, flp = greedy.finalize(tokens.reshape(1, 5, -1), sum_lp.reshape(1, 5))
    fix["upd_final_tokens"], fix["upd_final_sumlp"] = ftok.numpy(), np.array(flp, dtype=np.float64)

    # (2) ranker: lengths and sums chosen so that the three rules disagree
    groups_len = [[3, 9, 5], [12, 2, 7], [4, 4, 4], [1, 30, 10]]
    groups_lp = [[-2.0, -4.5, -3.1], [-9.0, -2.2, -6.5], [-1.0, -0.9, -1.1], [-0.8, -12.0, -5.0]]
    toks = [[torch.zeros(n, dtype=torch.long) for n in g] for g in groups_len]
    fix["rank_lengths"], fix["rank_sumlp"] = np.array(groups_len), np.array(groups_lp)
    for tag, pen in (("none", None), ("0p6", 0.6), ("1p0", 1.0)):
        fix[f"rank_{tag}"] = np.array([int(i) for i in dec.MaximumLikelihoodRanker(pen).rank(toks, groups_lp)])

    # (3) the loop itself
    sup = sorted(set(list(tk.non_speech_tokens) + [tk.transcribe, tk.translate, tk.sot, tk.sot_prev, tk.sot_lm, tk.no_speech]))
    ftk = _Tok(ids, tk.blank_tokens())
    ftk.no_speech = ids.no_speech
    ftk.decode = lambda t: " ".join(str(int(x)) for x in t)
    n_audio, n_group, sample_len = 2, 3, 12
    for tag, pen in (("none", None), ("0p6", 0.6)):
        calls = []

        def decode(x, cross, past, calls=calls):
            calls.append(tuple(x.shape))
            return torch.from_numpy(DR.sampling_logits(len(calls) - 1, x.shape[0], x.shape[1])), None
        me = types.SimpleNamespace(
            tokens=torch.tensor([[ids.sot, ids.lang0, ids.transcribe]] * n_audio), n_group=n_group, sample_len=sample_len,
            initial_token_length=3, sot_index=0, sample_begin=3, tokenizer=ftk,
            decoder_config={"num_text_ctx": 448, "num_audio": n_audio},
            logit_filters=[dec.SuppressBlank(ftk, 3), dec.SuppressTokens(sup), dec.ApplyTimestampRules(ftk, 3, 50)],
            decoder=dec.GreedyDecoder(0.7, ids.eot), sequence_ranker=dec.MaximumLikelihoodRanker(pen),
            options=types.SimpleNamespace(temperature=0.7), xa2cross_key_value=lambda xa: None, decode=decode,
            compression_ratio=lambda text: 1.0)
        xa = torch.zeros(n_audio * n_group, 1, 1)             # the reference expects one feature row per candidate (:829)
        torch.manual_seed(99)
        tokens, sum_lp, nsp = dec.WhisperDecoding.main_loop(me, xa)
        res = dec.WhisperDecoding.post_process(me, tokens, sum_lp, nsp, xa, ["en"] * n_audio)
        fix[f"loop_{tag}_tokens"] = tokens.numpy()
        fix[f"loop_{tag}_sumlp"] = sum_lp.numpy()
        fix[f"loop_{tag}_nsp"] = np.array(nsp, dtype=np.float64)
        fix[f"loop_{tag}_calls"] = np.array(calls)
        fix[f"loop_{tag}_selected_tokens"] = np.array([" ".join(map(str, r.tokens)) for r in res])
        fix[f"loop_{tag}_avg_logprob"] = np.array([r.avg_logprob for r in res], dtype=np.float64)
        fix[f"loop_{tag}_nsp_selected"] = np.array([r.no_speech_prob for r in res], dtype=np.float64)
    fix["suppress"] = np.array(sup)
    np.savez_compressed(os.path.join(OUT, "sampling.npz"), **fix)
    print("sampling.npz: update next", [fix[f"upd{s}_next"].tolist() for s in range(3)], "ranks", fix["rank_none"].tolist(),
          fix["rank_0p6"].tolist(), fix["rank_1p0"].tolist())
    print("  loop tokens", fix["loop_none_tokens"].shape, "selected", fix["loop_none_selected_tokens"].tolist(),
          fix["loop_0p6_selected_tokens"].tolist())


def gen_mel_fixture():
    """log_mel_spectrogram(seeded noise) from the reference's own whisper_utils (next-scope row f1)."""
    sys.path.insert(0, W)
    import whisper_utils as wu
    rng = np.random.Generator(np.random.Philox(99))
    audio = (rng.standard_normal(16000 * 2) * 0.1).astype(np.float32)
    mel = wu.log_mel_spectrogram(wu.pad_or_trim(audio, 16000 * 3))
    filt = wu.mel_filters("cpu", 80).numpy()
    np.savez_compressed(os.path.join(OUT, "mel.npz"), audio_seed=99, n_audio=16000 * 2, n_padded=16000 * 3,
                        mel=mel.numpy().astype(np.float32), filters_checksum=float(np.abs(filt).sum()),
                        filters_rowsum=filt.sum(axis=1).astype(np.float32))
    print("mel.npz written", tuple(mel.shape))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    tm = import_reference_model()
    if "--only-tiny" in sys.argv:
        gen_tiny_en_shape_fixture(tm)
        sys.exit(0)
    if "--only-rules-nots" in sys.argv:
        gen_rules_nots_fixture(import_reference_decoding())
        sys.exit(0)
    if "--only-ops" in sys.argv:
        gen_op_fixtures(tm)
        sys.exit(0)
    if "--only-sampling" in sys.argv:
        gen_sampling_fixture(import_reference_decoding())
        sys.exit(0)
    gen_model_fixture(tm)
    gen_tiny_en_shape_fixture(tm)
    gen_op_fixtures(tm)
    gen_mel_fixture()
    dec = import_reference_decoding()
    gen_rules_fixture(dec)
    gen_rules_nots_fixture(dec)
    gen_sampling_fixture(dec)
