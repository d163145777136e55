"""The oracle (oracle/*.py) against the golden vectors produced by the reference itself
(oracle/gen_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import decoding_rules as DR
from oracle.whisper_oracle import (Dims, OracleConfig, OracleModel, greedy_reference_run,
                                   synthetic_mel, synthetic_state_dict, kv_quantize,
                                   symmetric_quantize_int8, dequantize_int8, woq_reference_matmul,
                                   woq_colwise_atol, _gelu)


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "model_micro.npz"))


def _dims(fx):
    return Dims(**{k: int(v) for k, v in zip(fx["dims_keys"], fx["dims"])})


def _run(fx, act):
    dims = _dims(fx)
    sd = synthetic_state_dict(dims, int(fx["seed"]))
    m = OracleModel(dims, sd, OracleConfig(act=act))
    mel = synthetic_mel(int(fx["batch"]), 2 * dims.n_audio_ctx, dims.n_mels, int(fx["mel_seed"]))
    return dims, greedy_reference_run(m, mel, fx["prompt"].tolist(), int(fx["n_steps"]))


def _heads_to_flat(t):   # [B,H,T,64] -> [B,T,C]
    B, H, T, D = t.shape
    return t.permute(0, 2, 1, 3).reshape(B, T, H * D)


def test_fp32_model_matches_reference(fx):
    dims, r = _run(fx, "float32")
    np.testing.assert_allclose(r["xa"].numpy(), fx["f32_xa"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(_heads_to_flat(r["cross_kv"][0][:, 0]).numpy(), fx["f32_cross_k0"], atol=2e-5)
    np.testing.assert_allclose(_heads_to_flat(r["cross_kv"][0][:, 1]).numpy(), fx["f32_cross_v0"], atol=2e-5)
    np.testing.assert_allclose(_heads_to_flat(r["cross_kv"][-1][:, 1]).numpy(), fx["f32_cross_vL"], atol=2e-5)
    np.testing.assert_allclose(r["logits"][0].numpy(), fx["f32_prefill_logits"], atol=5e-5)
    steps = np.stack([l[:, 0].numpy() for l in r["logits"][1:]], axis=1)
    np.testing.assert_allclose(steps, fx["f32_step_logits"], atol=5e-5)
    assert r["ids"].tolist() == fx["f32_ids"].tolist()
    np.testing.assert_allclose(_heads_to_flat(r["self_kv"][0][:, 0]).numpy(), fx["f32_self_k0"], atol=2e-5)
    np.testing.assert_allclose(_heads_to_flat(r["self_kv"][-1][:, 1]).numpy(), fx["f32_self_vL"], atol=2e-5)


def test_fp16_model_matches_reference_fp16_mode(fx):
    """fp16 mode: the reference ran torch's CPU half kernels, the oracle rounds explicitly; they
    agree to a few fp16 ulps of the logit scale and on every greedy id."""
    dims, r = _run(fx, "float16")
    assert np.abs(r["xa"].numpy() - fx["f16_xa"]).max() < 2e-2
    d = np.abs(r["logits"][0].numpy() - fx["f16_prefill_logits"]).max()
    assert d < 3e-2, d
    steps = np.stack([l[:, 0].numpy() for l in r["logits"][1:]], axis=1)
    assert np.abs(steps - fx["f16_step_logits"]).max() < 3e-2
    assert r["ids"].tolist() == fx["f16_ids"].tolist()
    # and fp16 mode stays close to fp32 mode (tolerance accounting for the GPU tests)
    assert np.abs(steps - fx["f32_step_logits"]).max() < 5e-2


def test_ops_match_reference(golden_dir):
    ops = np.load(os.path.join(golden_dir, "ops.npz"))
    dims = Dims(80, 9, 128, 2, 0, 8, 4, 128, 2, 0)
    m = OracleModel(dims, {}, OracleConfig(act="float32"))
    q, k, v = (torch.from_numpy(ops[n]) for n in ("attn_q", "attn_k", "attn_v"))
    np.testing.assert_allclose(m._attend(q, k, v, 2).numpy(), ops["attn_out"], atol=2e-6)
    mask = torch.full((9, 9), float("-inf")).triu_(1)
    np.testing.assert_allclose(m._attend(k, k, v, 2, mask).numpy(), ops["attn_causal_out"], atol=2e-6)
    m.p["ln.weight"], m.p["ln.bias"] = torch.from_numpy(ops["ln_w"]), torch.from_numpy(ops["ln_b"])
    np.testing.assert_allclose(m._ln(torch.from_numpy(ops["ln_x"]), "ln").numpy(), ops["ln_out"], atol=2e-6)
    m16 = OracleModel(dims, {}, OracleConfig(act="float16"))
    m16.p = m.p
    x16 = torch.from_numpy(ops["ln_x"]).half().float()
    np.testing.assert_array_equal(m16._ln(x16, "ln").numpy(), ops["ln_out_f16"])
    np.testing.assert_allclose(_gelu(torch.from_numpy(ops["gelu_x"]), "erf").numpy(), ops["gelu_out"], atol=1e-6)
    # tanh GELU stays within the tolerance the reference's own test accepts (test_gelu.py:44-48)
    np.testing.assert_allclose(_gelu(torch.from_numpy(ops["gelu_x"]), "tanh").numpy(), ops["gelu_out"], atol=1e-3)
    x = torch.from_numpy(ops["conv_x"])
    y1 = torch.nn.functional.gelu(torch.nn.functional.conv1d(
        x, torch.from_numpy(ops["conv1_w"]), torch.from_numpy(ops["conv1_b"]), padding=1))
    np.testing.assert_allclose(y1.numpy(), ops["conv1_out"], atol=1e-6)


def _rules(fixr):
    return DR.RuleSet(DR.MULTILINGUAL, 3, fixr["suppress"].tolist(), fixr["blank"].tolist(), 50)


def test_decoding_rules_match_reference(golden_dir):
    fixr = np.load(os.path.join(golden_dir, "decoding_rules.npz"))
    rules = _rules(fixr)
    cases = DR.golden_rule_cases()
    assert len(cases) == int(fixr["n_cases"])
    for c, (toks, logits) in enumerate(cases):
        assert abs(float(np.abs(logits).sum(dtype=np.float64)) - float(fixr[f"c{c}_logits_checksum"])) < 1e-6
        np.testing.assert_array_equal(toks, fixr[f"c{c}_tokens"])
        lg = DR.apply_filters(logits[None], toks[None], rules)
        isinf = np.unpackbits(fixr[f"c{c}_filtered_isinf"])[:lg.shape[1]].astype(bool)
        np.testing.assert_array_equal(np.isinf(lg[0]), isinf, err_msg=f"case {c}")
        s = np.zeros(1, dtype=np.float32)
        new_tokens, done = DR.greedy_update(toks[None], lg, s, rules.ids.eot)
        assert int(new_tokens[0, -1]) == int(fixr[f"c{c}_next"]), c
        assert abs(float(s[0]) - float(fixr[f"c{c}_sumlp"])) < 1e-4, c
        assert done == bool(fixr[f"c{c}_done"])


def test_decoding_rules_without_timestamps_match_reference(golden_dir):
    """DecodingOptions.without_timestamps: the reference builds SuppressBlank + SuppressTokens and no ApplyTimestampRules
    (W/decoding.py:332-348); tests/golden/decoding_rules_nots.npz holds what its two filter classes and GreedyDecoder.update give
    on the rule cases with <|notimestamps|> in the start sequence (oracle/gen_golden.py: gen_rules_nots_fixture)."""
    import dataclasses
    fixr, fixn = np.load(os.path.join(golden_dir, "decoding_rules.npz")), np.load(os.path.join(golden_dir, "decoding_rules_nots.npz"))
    rules = dataclasses.replace(_rules(fixr), sample_begin=4, timestamps=False)
    cases = DR.golden_rule_cases()
    assert len(cases) == int(fixn["n_cases"])
    differ = 0
    for c, (toks, logits) in enumerate(cases):
        toks = np.concatenate([toks[:3], [rules.ids.no_timestamps], toks[3:]])
        lg = DR.apply_filters(logits[None], toks[None], rules)
        s = np.zeros(1, dtype=np.float32)
        new_tokens, done = DR.greedy_update(toks[None], lg, s, rules.ids.eot)
        assert int(new_tokens[0, -1]) == int(fixn[f"c{c}_next"]), c
        assert abs(float(s[0]) - float(fixn[f"c{c}_sumlp"])) < 1e-4, c
        assert done == bool(fixn[f"c{c}_done"])
        differ += int(fixn[f"c{c}_next"]) != int(fixr[f"c{c}_next"])
    assert differ >= 12                    # not the timestamp rules' answers: the mode is really exercised


def test_special_ids():
    ids = DR.MULTILINGUAL
    assert (ids.eot, ids.sot, ids.lang0, ids.translate, ids.transcribe, ids.sot_lm, ids.sot_prev,
            ids.no_speech, ids.no_timestamps, ids.timestamp_begin, ids.n_vocab) == \
        (50257, 50258, 50259, 50358, 50359, 50360, 50361, 50362, 50363, 50364, 51865)


# ---- quantisation known-answer tests restated from the reference's own tests -----------------

def _woq_gen(n, k, seed=0):
    """R/tests/quantization/_utils.py:15-23: uniform(-1, 1) fp16 weights."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand((n, k), generator=g, dtype=torch.float32) * 2 - 1).half()


@pytest.mark.parametrize("m,n,k", [(1, 1024, 4096), (16, 768, 1536)])
def test_weight_only_quant_matmul_spec(m, n, k):
    """test_weight_only_quant_matmul.py:94-119 with the plugin replaced by the restated kernel
    arithmetic (dequantise per element, fp32 accumulate, fp16 store)."""
    x = (_woq_gen(m, k, 1) * 200.0)
    w_kn = _woq_gen(k, n, 2)                        # [K, N] like the reference test
    q, s = symmetric_quantize_int8(w_kn.t().contiguous().numpy())     # per output channel
    ref = woq_reference_matmul(x.numpy(), q.T, s)
    wdq = dequantize_int8(q, s).astype(np.float32)             # [N, K]
    act = (x.float().numpy() @ wdq.T).astype(np.float16)
    atol = woq_colwise_atol(ref)
    assert (np.abs(act.astype(np.float32) - ref.astype(np.float32)) <= atol[None, :] + 1e-3).all()


def test_symmetric_quantize_properties():
    w = _woq_gen(64, 256, 3).numpy()
    w[5] = 0                                  # all-zero channel
    w[7, 3] = np.float16(0.75)
    q, s = symmetric_quantize_int8(w)
    assert q.dtype == np.int8 and s.dtype == np.float16
    assert (q[5] == 0).all() and s[5] == 0
    absmax = np.abs(w.astype(np.float32)).max(axis=1)
    np.testing.assert_array_equal(s, (absmax / 128).astype(np.float16))
    # the channel maximum always maps to +-128 -> clipped to 127 on the positive side
    for r in (0, 7, 11):
        j = np.abs(w[r].astype(np.float32)).argmax()
        assert q[r, j] in (127, -128)
    # round half away from zero: w/scale == 0.5 exactly -> 1
    w2 = np.zeros((1, 4), dtype=np.float16)
    w2[0] = [1.0, 1.0 / 256, -1.0 / 256, 3.0 / 256]
    q2, _ = symmetric_quantize_int8(w2)
    assert q2.tolist() == [[127, 1, -1, 2]]
    # reconstruction error bound
    err = np.abs(dequantize_int8(q, s).astype(np.float32) - w.astype(np.float32))
    assert (err <= (absmax / 128)[:, None] * 1.15 + 1e-6).all()   # +fp16 rounding of scale and product


def test_quantize_tensor_spec():
    """test_functional.py:22-50: (x * s).round().clip(-128, 127) exactly, RNE."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 2, 2, 4), generator=g).half().float()
    got = kv_quantize(x, 1.0 / 0.4)
    want = (x * np.float32(1.0) / np.float32(1.0 / 0.4)).round().clip(-128, 127).to(torch.int8)
    # 1/(1/0.4) in fp32 is what the engine multiplies by
    inv = float(np.float32(1.0) / np.float32(1.0 / 0.4))
    want = (x * inv).round().clip(-128, 127).to(torch.int8)
    assert torch.equal(got, want)
    assert kv_quantize(torch.tensor([0.5, 1.5, 2.5, -0.5, 1000.0, -1000.0]), 1.0).tolist() == [0, 2, 2, 0, 127, -128]


def test_int8_kv_requantisation_is_idempotent():
    """The reference re-quantises dequant(past) every step (attention.py:296-348); an in-place
    append is only a valid restatement if that round trip is the identity."""
    q = torch.arange(-128, 128, dtype=torch.int8)
    for t in (1e-4, 3.7e-3, 0.0123, 0.5):
        from oracle.whisper_oracle import kv_dequantize
        assert torch.equal(kv_quantize(kv_dequantize(q, t, "float16"), t), q)


def test_int8_kv_and_weight_only_model_close_to_fp16(fx):
    dims = _dims(fx)
    sd = synthetic_state_dict(dims, int(fx["seed"]))
    mel = synthetic_mel(2, 2 * dims.n_audio_ctx, dims.n_mels, int(fx["mel_seed"]))
    base = OracleModel(dims, sd, OracleConfig(act="float16"))
    scales = base.calibrate_kv_scales(mel, 6)
    assert len(scales) == dims.n_text_layer and all(s > 0 for s in scales)
    r0 = greedy_reference_run(base, mel, fx["prompt"].tolist(), 6)
    mq = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=True, int8_kv=True, kv_scales=scales))
    r1 = greedy_reference_run(mq, mel, fx["prompt"].tolist(), 6)
    assert r1["self_kv"][0].dtype == torch.int8
    d = max(float((a - b).abs().max()) for a, b in zip(r0["logits"], r1["logits"]))
    assert d < 0.5, d          # quantisation noise, not garbage (logit std is ~1.5)


# ---- a REAL Whisper shape (tiny.en width / heads / 1500 audio positions / gpt2 vocabulary), from the reference ---------
@pytest.fixture(scope="module")
def fx_tiny(golden_dir):
    return np.load(os.path.join(golden_dir, "model_tiny_en_shape.npz"))


def _run_tiny(fx, act):
    dims = _dims(fx)
    sd = synthetic_state_dict(dims, int(fx["seed"]))
    mel = synthetic_mel(int(fx["batch"]), 2 * dims.n_audio_ctx, dims.n_mels, int(fx["mel_seed"]))
    with torch.no_grad():
        r = greedy_reference_run(OracleModel(dims, sd, OracleConfig(act=act)), mel, fx["prompt"].tolist(), int(fx["n_steps"]))
    last = np.stack([l[:, -1].numpy() for l in r["logits"]], axis=1)            # [B, n_steps, V]
    return dims, r, last


@pytest.mark.parametrize("tag,act,tol_x,tol_l", [("f32", "float32", 1e-4, 2e-4), ("f16", "float16", 2e-2, 3e-2)])
def test_tiny_en_shape_matches_reference(fx_tiny, tag, act, tol_x, tol_l):
    """1500 keys, 6 heads, 384 wide, 51 864 tokens: the shapes the MICRO fixture cannot reach.  The reference
    (W/torch_model.py) produced the rows / top-64 logits stored here; the oracle must reproduce them."""
    fx = fx_tiny
    dims, r, last = _run_tiny(fx, act)
    assert (dims.n_audio_ctx, dims.n_audio_head, dims.n_audio_state, dims.n_vocab) == (1500, 6, 384, 51864)
    rows = fx["rows"]
    assert np.abs(r["xa"].numpy()[:, rows] - fx[f"{tag}_xa"].astype(np.float32)).max() < tol_x
    for key, layer, kv in (("cross_k0", 0, 0), ("cross_v0", 0, 1), ("cross_vL", -1, 1)):
        got = _heads_to_flat(r["cross_kv"][layer][:, kv]).numpy()[:, rows]
        assert np.abs(got - fx[f"{tag}_{key}"].astype(np.float32)).max() < tol_x, key
    top = fx[f"{tag}_top_ids"].astype(np.int64)
    got_top = np.take_along_axis(last, top, axis=-1)
    assert np.abs(got_top - fx[f"{tag}_top_logits"]).max() < tol_l
    # nothing outside the stored top-64 overtakes them, and the whole row agrees in aggregate
    assert (np.sort(last, axis=-1)[..., -64:].min(-1) >= fx[f"{tag}_top_logits"].min(-1) - tol_l).all()
    np.testing.assert_allclose(np.abs(last).sum(-1, dtype=np.float64), fx[f"{tag}_logit_checksum"], rtol=2e-4 if tag == "f32" else 5e-3)
    assert r["ids"].tolist() == fx[f"{tag}_ids"].tolist()
    assert float(fx[f"{tag}_margins"].min()) > 2 * tol_l          # the ids are not near-ties: the equality above is meaningful
