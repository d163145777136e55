"""The example's PyTorch comparison path (BASELINE.json configs[0]: Whisper tiny.en through PyTorch on the host CPU) on CPU:

  * the package's own PyTorch Whisper (`torch_model.py`, a functional model over the checkpoint's flat state dict) against the
    outputs the REFERENCE's model produced in the build container (tests/golden/model_micro.npz, model_tiny_en_shape.npz;
    W/torch_model.py run by oracle/gen_golden.py), fp32 and the reference's fp16-input mode;
  * `WhisperEncoding.torch_get_audio_features`, `WhisperDecoding.torch_detect_language`, `torch_main_loop`, `post_process`
    (W/encoding.py:43-46, W/decoding.py:661-701,743-783,827-878) driven through the product API at the tiny.en-shaped fixture,
    with the package's model AND with a duck-typed wrapper around the oracle (`.encoder`, `.logits`, `.decoder(.., kv_cache=)`,
    `install_kv_cache_hooks`): two independent implementations, one loop -- and the oracle's restated rules
    (oracle/decoding_rules.py, pinned to the reference's filter classes) around the oracle model as the third;
  * `summarize.py --test_torch` end to end with `--torch_model` naming any object of that interface.
"""
import os

import numpy as np
import pytest
import torch

import build as B
import synthetic
import torch_model as TM
from decoding import DecodingOptions, WhisperDecoding
from encoding import WhisperEncoding
from oracle import decoding_rules as DR
from oracle.whisper_oracle import Dims, OracleConfig, OracleModel, synthetic_mel, synthetic_state_dict

torch.set_num_threads(min(8, os.cpu_count() or 1))


def _fixture_model(fx):
    dims = Dims(*[int(v) for v in fx["dims"]])
    sd = synthetic_state_dict(dims, int(fx["seed"]))
    mel = synthetic_mel(int(fx["batch"]), 2 * dims.n_audio_ctx, dims.n_mels, int(fx["mel_seed"]))
    model = TM.Whisper(TM.ModelDimensions(**dims.to_dict())).load_state_dict({k: v.float() for k, v in sd.items()})
    return dims, sd, mel, model


def _run_like_the_generator(model, mel, prompt, n_steps, half):
    """The protocol oracle/gen_golden.py: run_reference_model drove the reference's model with."""
    x = mel.half() if half else mel.float()
    with torch.no_grad():
        xa = model.encoder(x)
        cache, hooks = model.install_kv_cache_hooks()
        cur = torch.tensor([prompt] * mel.shape[0])
        logits_all, ids = [], []
        for _ in range(n_steps):
            logits = model.decoder(cur, xa, kv_cache=cache)
            assert logits.dtype == torch.float32
            nxt = logits[:, -1].argmax(-1)
            logits_all.append(logits.numpy())
            ids.append(nxt.numpy())
            cur = nxt[:, None]
        for h in hooks:
            h.remove()
    return xa, cache, logits_all, np.stack(ids, axis=1)


@pytest.mark.parametrize("tag,half,tol_x,tol_l", [("f32", False, 1e-4, 2e-4), ("f16", True, 2e-2, 3e-2)])
def test_package_torch_model_matches_the_reference_model_micro(golden_dir, tag, half, tol_x, tol_l):
    fx = np.load(os.path.join(golden_dir, "model_micro.npz"))
    dims, sd, mel, model = _fixture_model(fx)
    xa, cache, logits_all, ids = _run_like_the_generator(model, mel, fx["prompt"].tolist(), int(fx["n_steps"]), half)
    assert xa.dtype == (torch.float16 if half else torch.float32)
    assert np.abs(xa.float().numpy() - fx[f"{tag}_xa"]).max() < tol_x
    last = dims.n_text_layer - 1
    for key, name in (("cross_k0", "dec.0.cross.k"), ("cross_v0", "dec.0.cross.v"), ("cross_vL", f"dec.{last}.cross.v"),
                      ("self_k0", "dec.0.self.k"), ("self_vL", f"dec.{last}.self.v")):
        assert np.abs(cache[name].float().numpy() - fx[f"{tag}_{key}"]).max() < tol_x, key
    assert np.abs(logits_all[0] - fx[f"{tag}_prefill_logits"]).max() < tol_l
    steps = np.stack([l[:, 0] for l in logits_all[1:]], axis=1)
    assert np.abs(steps - fx[f"{tag}_step_logits"]).max() < tol_l
    safe = fx[f"{tag}_margins"] > 2 * tol_l
    assert (ids[safe] == fx[f"{tag}_ids"][safe]).all() and safe.sum() >= ids.size - 2


@pytest.mark.parametrize("tag,half,tol_x,tol_l", [("f32", False, 1e-4, 2e-4), ("f16", True, 2e-2, 3e-2)])
def test_package_torch_model_matches_the_reference_model_tiny_en_shape(golden_dir, tag, half, tol_x, tol_l):
    fx = np.load(os.path.join(golden_dir, "model_tiny_en_shape.npz"))
    dims, sd, mel, model = _fixture_model(fx)
    assert (dims.n_audio_ctx, dims.n_audio_head, dims.n_audio_state, dims.n_vocab) == (1500, 6, 384, 51864) and not model.is_multilingual
    xa, cache, logits_all, ids = _run_like_the_generator(model, mel, fx["prompt"].tolist(), int(fx["n_steps"]), half)
    rows = fx["rows"]
    assert np.abs(xa.float().numpy()[:, rows] - fx[f"{tag}_xa"].astype(np.float32)).max() < tol_x
    for key, name in (("cross_k0", "dec.0.cross.k"), ("cross_v0", "dec.0.cross.v"), ("cross_vL", f"dec.{dims.n_text_layer - 1}.cross.v")):
        assert np.abs(cache[name].float().numpy()[:, rows] - fx[f"{tag}_{key}"].astype(np.float32)).max() < tol_x, key
    last = np.stack([l[:, -1] for l in logits_all], axis=1)                      # [B, n_steps, V]
    top = fx[f"{tag}_top_ids"].astype(np.int64)
    assert np.abs(np.take_along_axis(last, top, axis=-1) - fx[f"{tag}_top_logits"]).max() < tol_l
    np.testing.assert_allclose(np.abs(last).sum(-1, dtype=np.float64), fx[f"{tag}_logit_checksum"], rtol=2e-4 if tag == "f32" else 5e-3)
    assert ids.tolist() == fx[f"{tag}_ids"].tolist()
    assert float(fx[f"{tag}_margins"].min()) > 2 * tol_l


# ---- the wrappers' torch_* entry points --------------------------------------------------------------------------------------
class OracleAsTorchModel:
    """The oracle behind the interface the wrappers drive (tests only)."""
    def __init__(self, oracle):
        self.o = oracle

    def encoder(self, mel):
        return self.o.encoder(mel)

    def logits(self, tokens, xa):
        return self.o.decoder(tokens, self.o.cross_kv(xa), None)[0]

    def decoder(self, tokens, xa, kv_cache=None):
        if kv_cache is None:
            return self.logits(tokens, xa)
        if "ckv" not in kv_cache:
            kv_cache["ckv"], kv_cache["kv"] = self.o.cross_kv(xa), None
        logits, kv_cache["kv"] = self.o.decoder(tokens, kv_cache["ckv"], kv_cache["kv"])
        return logits

    def install_kv_cache_hooks(self):
        return {}, []


def _engine_dir(tmp, dims: Dims, sd):
    out = tmp / "eng"
    args = B.parse_arguments(["--output_dir", str(out), "--use_gpt_attention_plugin", "--use_gemm_plugin", "--use_layernorm_plugin", "--log_level", "error"])
    B.build_from_checkpoint({"dims": dims.to_dict(), "model_state_dict": sd}, args)
    return out


@pytest.mark.parametrize("half", [False, True])
def test_torch_entry_points_through_the_product_api_tiny_en_shape(golden_dir, tmp_path, half):
    """tiny.en's shape (English-only vocabulary: `torch_detect_language` answers 'en' without a pass, W/decoding.py:661-663 with
    is_multilingual False): audio features against the reference's golden rows, then `torch_main_loop` -- Whisper's logit rules on
    the host around `model.decoder(.., kv_cache=)` -- with two independent models and the oracle's restated loop, `post_process`."""
    fx = np.load(os.path.join(golden_dir, "model_tiny_en_shape.npz"))
    dims, sd, mel, model = _fixture_model(fx)
    tag = "f16" if half else "f32"
    eng = _engine_dir(tmp_path, dims, sd)
    enc = WhisperEncoding(eng, only_torch=True)
    dec = WhisperDecoding(eng, only_torch=True, options=DecodingOptions(sample_len=6))
    assert not dec.is_multilingual and dec.initial_tokens == (50257,) and dec.tokenizer.eot == 50256
    x = mel.half() if half else mel.float()
    xa = enc.torch_get_audio_features(model, x)
    assert np.abs(xa.float().numpy()[:, fx["rows"]] - fx[f"{tag}_xa"].astype(np.float32)).max() < (2e-2 if half else 1e-4)
    languages, probs = dec.torch_detect_language(model, xa)
    assert languages == ["en"] and probs is None
    tokens, sum_lp, nsp = dec.torch_main_loop(model, xa)
    assert dec.kv_cache == {} and dec.hooks == []                               # the loop hands its cache back
    oracle = OracleModel(dims, sd, OracleConfig(act="float16" if half else "float32"))
    duck = OracleAsTorchModel(oracle)
    xa_o = enc.torch_get_audio_features(duck, x)
    tokens_o, sum_lp_o, nsp_o = dec.torch_main_loop(duck, xa_o)
    # the third: the oracle's restatement of the reference's rules (pinned to its filter classes by tests/golden/decoding_rules*.npz)
    rules = DR.RuleSet(DR.SpecialIds(50256), dec.sample_begin, list(dec._get_suppress_tokens()), list(dec.tokenizer.blank_tokens()) + [dec.tokenizer.eot],
                       dec.max_initial_timestamp_index)
    state = {"kv": None}
    ckv = oracle.cross_kv(oracle.encoder(x))

    def step(feed, first):
        logits, state["kv"] = oracle.decoder(torch.from_numpy(feed), ckv, None if first else state["kv"])
        return logits.numpy()
    tokens_r, sum_lp_r, nsp_r = DR.main_loop(step, np.array([list(dec.initial_tokens)] * mel.shape[0], dtype=np.int64), rules, 6, dims.n_text_ctx)
    assert tokens_o.tolist() == tokens_r.tolist()
    assert tokens.tolist() == tokens_r.tolist()
    tol = 6 * (3e-2 if half else 2e-4)
    assert np.allclose(sum_lp.numpy(), sum_lp_r, atol=tol) and np.allclose(sum_lp_o.numpy(), sum_lp_r, atol=tol)
    assert np.allclose(nsp, nsp_r, atol=1e-3) and np.allclose(nsp_o, nsp_r, atol=1e-3)
    assert tokens.shape[1] == 1 + 6 and len(set(tokens[0, 1:].tolist())) > 2
    res = dec.post_process(tokens, sum_lp, nsp, xa, languages)
    assert len(res) == 1 and res[0].language == "en" and res[0].tokens == [t for t in tokens[0, 1:].tolist() if t != dec.tokenizer.eot][:len(res[0].tokens)]
    assert isinstance(res[0].text, str) and res[0].no_speech_prob == pytest.approx(nsp[0])


def test_torch_detect_language_multilingual_two_models_agree(tmp_path):
    """A multilingual vocabulary (51 865 tokens): `torch_detect_language` runs `model.logits([[sot]], audio_features)`, masks everything
    but the 99 language tokens and writes the winner into the start sequence (W/decoding.py:661-701) -- the package's model and the
    oracle behind the same interface agree on the language, its probability table and the start sequence."""
    dims = Dims(**synthetic.DIMS["micro-fullvocab"])
    sd = synthetic_state_dict(dims, 5)
    mel = synthetic_mel(2, 2 * dims.n_audio_ctx, dims.n_mels, 77).float()
    eng = _engine_dir(tmp_path, dims, sd)
    enc = WhisperEncoding(eng, only_torch=True)
    model = TM.Whisper(TM.ModelDimensions(**dims.to_dict())).load_state_dict({k: v.float() for k, v in sd.items()})
    duck = OracleAsTorchModel(OracleModel(dims, sd, OracleConfig(act="float32")))
    got = []
    for m in (model, duck):
        dec = WhisperDecoding(eng, only_torch=True, options=DecodingOptions(sample_len=4))
        assert dec.is_multilingual
        xa = enc.torch_get_audio_features(m, mel)
        languages, probs = dec.torch_detect_language(m, xa)
        tokens, sum_lp, nsp = dec.torch_main_loop(m, xa)
        got.append((languages, probs, dec.tokens.clone(), tokens, sum_lp))
    (l0, p0, s0, t0, lp0), (l1, p1, s1, t1, lp1) = got
    assert l0 == l1 and len(l0) == 2 and torch.equal(s0, s1) and int(s0[0, 1]) in range(50259, 50358)
    for a, b in zip(p0, p1):
        assert set(a) == set(b) and len(a) == 99 and max(abs(a[k] - b[k]) for k in a) < 1e-4 and abs(sum(a.values()) - 1.0) < 1e-3
    assert torch.equal(t0, t1) and torch.allclose(lp0, lp1, atol=1e-3)


def test_summarize_test_torch_accepts_any_model_of_the_interface(tmp_path, golden_dir, monkeypatch):
    """`summarize.py --test_torch` end to end on CPU (FLAC in, host log-mel, the PyTorch path, text clean-up, normaliser, WER), the
    model named by `--torch_model module:callable` -- any object with `.encoder`, `.logits`, `.decoder(.., kv_cache=)` and
    `install_kv_cache_hooks`; without the flag the package's own `torch_model.load_model(--checkpoint_file)` serves."""
    import shutil
    import sys
    import summarize as S
    dims = Dims(80, 1500, 64, 2, 1, 51864, 448, 64, 2, 1)                      # tiny.en's interface, a toy's width
    sd = synthetic_state_dict(dims, 3)
    eng = _engine_dir(tmp_path, dims, sd)
    ck = tmp_path / "toy.pt"
    torch.save({"dims": dims.to_dict(), "model_state_dict": sd}, ck)
    chapter = tmp_path / "ds" / "1089" / "134691"
    chapter.mkdir(parents=True)
    for i in range(2):
        shutil.copy(os.path.join(golden_dir, "librispeech_1089-134691-0000.flac"), chapter / f"1089-134691-000{i}.flac")
    (chapter / "1089-134691.trans.txt").write_text("".join(f"1089-134691-000{i} HE COULD WAIT NO LONGER\n" for i in range(2)))
    mod = tmp_path / "my_models.py"
    mod.write_text("import torch_model as TM\n"
                   "def make(checkpoint_file, device):\n"
                   "    return TM.load_model(checkpoint_file, device)\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    reports = []
    for extra in ([], ["--torch_model", "my_models:make"]):
        args = S.parse_arguments(["--test_torch", "--engine_dir", str(eng), "--dataset_dir", str(tmp_path / "ds"), "--checkpoint_file", str(ck),
                                  "--device", "cpu", "--sample_len", "4", "--log_level", "error"] + extra)
        reports.append(S.main(args)["Torch"])
    sys.modules.pop("my_models", None)
    assert reports[0]["utterances"] == 2 and reports[0]["hypotheses"] == reports[1]["hypotheses"]
    assert reports[0]["hypotheses"][0] == reports[0]["hypotheses"][1]          # the same clip twice: deterministic
    assert 0.0 <= reports[0]["wer"]
