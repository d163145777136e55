"""The reference disagrees with itself between M = 1 and M > 1 (VERDICT r5 "missing" 3): its weight-only plugin takes a GEMV kernel for one
activation row that rounds EVERY product to fp16 before the fp32 sum (weightOnlyMatrixVectorMultiplication.cu:187), and CUTLASS -- exact
fp16 x fp16 products -- for more (weightOnlyQuantMatmulPlugin.cpp:182-197, default_fpA_intB_traits.h:30-109).  The engine follows the
CUTLASS contract at every M (DESIGN.md section 2 "M = 1"); the oracle restates both (woq_gemv_reference, OracleConfig.gemv_fp16_products)
and these tests say how far apart they are: on the six Linear shapes of a large-v2 decoder layer, and through a whole model."""
import numpy as np
import pytest
import torch

from oracle.whisper_oracle import (Dims, OracleConfig, OracleModel, dequantize_int8, greedy_reference_run, symmetric_quantize_int8,
                                   synthetic_mel, synthetic_state_dict, woq_colwise_atol, woq_gemv_reference, woq_reference_matmul)

DECODER_SHAPES = [("qkv", 1280, 3840), ("out", 1280, 1280), ("cross q", 1280, 1280), ("cross out", 1280, 1280), ("mlp1", 1280, 5120),
                  ("mlp2", 5120, 1280)]      # (K, N) of one large-v2 decoder layer


def test_gemv_restatement_known_answer():
    """woq_gemv_reference against the kernel's arithmetic written out element by element (Python floats + explicit fp16 roundings, the
    additions in index order), on a case small enough to read: 2 outputs x 6 inputs."""
    x = np.array([0.5, -1.25, 3.0, 0.1, -0.3, 2.5], dtype=np.float16)
    q = np.array([[127, -128], [3, 77], [-45, 1], [100, -100], [0, 5], [-7, 64]], dtype=np.int8)     # [K, N]
    s = np.array([0.0123, 0.0077], dtype=np.float16)
    want = []
    for n in range(2):
        acc = np.float32(0)
        for k in range(6):
            w16 = np.float16(np.float16(q[k, n]) * s[n])
            acc = np.float32(acc + np.float32(np.float16(x[k] * w16)))
        want.append(np.float16(acc))
    got = woq_gemv_reference(x, q, s)
    assert got.shape == (1, 2) and got.dtype == np.float16
    assert np.array_equal(got[0], np.array(want, dtype=np.float16))


@pytest.mark.parametrize("name,K,N", DECODER_SHAPES)
def test_fp16_rounded_products_against_exact_products_on_the_decoder_shapes(name, K, N):
    """|GEMV kernel's arithmetic - CUTLASS contract| on one activation row: bounded by a few fp16 ulps of the output (K products rounded to
    fp16, each off by <= half an ulp of a product ~ 1 / sqrt(K) of the output's size, added with random signs), 20 x inside the tolerance the
    reference's own test accepts for this Linear (1.5 * max / 128, R/tests/quantization/_utils.py:66-88) -- and both are inside that
    tolerance of the test's ground truth (fp32 (x @ q) * scale)."""
    rng = np.random.default_rng(K + N)
    w = (rng.standard_normal((N, K)) * 2 / np.sqrt(K)).astype(np.float16)          # the synthetic engines' Linear weights (gain 2)
    worst = 0.0
    for trial in range(3):
        x = (rng.standard_normal((1, K)) * (0.5 + trial)).astype(np.float16)
        q, s = symmetric_quantize_int8(w)
        exact = (x.astype(np.float32) @ dequantize_int8(q, s).astype(np.float32).T).astype(np.float16)      # CUTLASS: exact products, fp32 sum
        gemv = woq_gemv_reference(x, q.T, s)
        truth = woq_reference_matmul(x, q.T, s).astype(np.float32)
        atol = float(woq_colwise_atol(truth)[0])
        d = np.abs(gemv.astype(np.float32) - exact.astype(np.float32))
        ulp = float(np.spacing(np.float16(np.abs(exact.astype(np.float32)).max())))
        assert d.max() <= 3 * ulp, (name, trial, d.max(), ulp)
        assert d.max() <= atol / 10, (name, trial, d.max(), atol)
        assert np.sqrt((d ** 2).mean()) <= ulp / 2
        for y in (exact, gemv):
            assert np.abs(y.astype(np.float32) - truth).max() <= atol
        worst = max(worst, d.max() / atol)
    assert worst < 0.1


def test_whole_model_under_either_contract():
    """The two contracts through a whole weight-only model (micro dims, prefill of 3 tokens = CUTLASS under both, then one-row steps):
    teacher-forced logits agree to well inside the engine-vs-oracle tolerance (3e-2), greedy ids are the same."""
    from synthetic import DIMS
    dims = Dims(**DIMS["micro"])
    sd = synthetic_state_dict(dims, 7)
    mel = synthetic_mel(1, 2 * dims.n_audio_ctx, dims.n_mels, 99)
    runs = {}
    for flag in (False, True):
        m = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=True, gemv_fp16_products=flag))
        runs[flag] = greedy_reference_run(m, mel, [5, 17, 900], 6)
    assert torch.equal(runs[False]["logits"][0], runs[True]["logits"][0])          # the 3-token prefill: M = 3, CUTLASS either way
    worst = max(float((a - b).abs().max()) for a, b in zip(runs[False]["logits"][1:], runs[True]["logits"][1:]))
    assert 0 < worst < 1e-2, worst                                                   # the contracts DO differ at M = 1 -- by this much
    assert torch.equal(runs[False]["ids"], runs[True]["ids"])
