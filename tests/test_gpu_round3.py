"""Round 3 on a real MI355X, all through the C ABI:
  * per-row completion (W/decoding.py:817-819 stops its single utterance at EOT): finished rows drop out of the attention
    kernels, groups without live rows are no longer stepped -- token ids identical to the loop that keeps every row;
  * the sampling path on the engine: log-probabilities of the sampled candidates against the oracle teacher-forced with them;
  * int8 KV cache codes: every code that differs from the oracle's is PROVEN to come from a value within fp16 noise of a
    rounding boundary (per element, no percentage budget);
  * fp16 and weight-only-int8 engines at FULL depth (32 + 32 layers) against the GPU-resident oracle;
  * bounds the advisor asked for (key-range splits of the cross-attention merge, widths of the fused small-batch Linear).
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import build as B  # noqa: E402
import native  # noqa: E402
import synthetic  # noqa: E402
from decoding import DecodingOptions, WhisperDecoding  # noqa: E402
from encoding import WhisperEncoding  # noqa: E402
from oracle import decoding_rules as DR  # noqa: E402
from oracle.whisper_oracle import (Dims, OracleConfig, OracleModel, greedy_reference_run, kv_quantize, synthetic_mel,
                                   synthetic_state_dict)  # noqa: E402
from test_gpu_model import LOGIT_TOL, LOGIT_TOL_INT8_KV, _write_kv_scales, build_engine  # noqa: E402


@pytest.fixture(scope="module")
def tmpdir_module(tmp_path_factory):
    return str(tmp_path_factory.mktemp("engines3"))


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return native.load_library()


def stream():
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------ per-row completion
@pytest.mark.parametrize("n", [1, 7, 64, 65, 192, 1000])
def test_step_finish_builds_the_live_list(lib, n):
    g = torch.Generator().manual_seed(n)
    done = (torch.rand(n, generator=g) < 0.4).to(torch.int32)
    for flags in (done, torch.zeros(n, dtype=torch.int32), torch.ones(n, dtype=torch.int32)):
        d = flags.cuda()
        live = torch.full((1 + n,), -7, dtype=torch.int32, device="cuda")
        counter = torch.tensor([41], dtype=torch.int32, device="cuda")
        native.check(lib.wm_step_finish(counter.data_ptr(), d.data_ptr(), n, live.data_ptr(), stream()))
        torch.cuda.synchronize()
        want = torch.nonzero(flags == 0)[:, 0].to(torch.int32)
        assert int(counter) == 42 and int(live[0]) == len(want)
        assert torch.equal(live[1:1 + len(want)].cpu(), want)
        assert bool((live[1 + len(want):] == -7).all())                      # nothing written past the list
    with pytest.raises(native.WmError, match="batch"):
        native.check(lib.wm_step_finish(None, d.data_ptr(), 1025, live.data_ptr(), stream()))


@pytest.mark.parametrize("small_rows", [0, 8])
@pytest.mark.parametrize("batch", [2, 4, 6, 8, 40])
def test_decoder_step_with_live_rows(tmpdir_module, lib, batch, small_rows):
    """One decode step with a list of live rows: their logits and cache rows are bit-identical to the step that runs every
    row (a row's result never depends on which other rows are live), the other rows' KV caches are not touched.  Batch 6
    runs the key-split cross-attention + merge (and, with small_rows = 8, the fused small-batch path), batch 40 the
    one-workgroup-per-head kernel.  Round 5: with small_rows = 8 the batches up to 8 take the ONE-LAUNCH step (2, 4 and 6 / 8 rows: its three
    multi-row kernels), whose attention stages skip the finished rows the same way."""
    prev = lib.wm_set_small_batch_rows(small_rows)
    try:
        dims = Dims(**synthetic.DIMS["micro"])
        eng = build_engine(tmpdir_module, "micro", 7, True, True, [0.05, 0.06])
        enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
        mel = synthetic_mel(batch, 2 * dims.n_audio_ctx, dims.n_mels, 5).cuda()
        cross = dec.xa2cross_key_value(enc.get_audio_features(mel))
        cap, H, V = dims.n_text_ctx, dims.n_text_head, dims.n_vocab
        sess, pos = dec.decoder_session, dec.positional_embedding
        g = torch.Generator().manual_seed(batch)
        toks = torch.randint(0, V, (batch, 4), generator=g).to(torch.int32).cuda()

        def run(live):
            kv = [torch.zeros((batch, 2, H, cap, 64), dtype=torch.int8, device="cuda") for _ in range(dims.n_text_layer)]
            lg = torch.zeros((batch, 3, V), dtype=torch.float16, device="cuda")
            sess.decoder_step(toks[:, :3].contiguous(), pos[0:3], cross, None, cap, kv, cap, lg, 0, stream())
            marker = [t.clone() for t in kv]
            lg1 = torch.full((batch, 1, V), 123.0, dtype=torch.float16, device="cuda")
            sess.decoder_step(toks[:, 3:4].contiguous(), pos[3:4], cross, kv, cap, kv, cap, lg1, 3, stream(), live_rows=live)
            torch.cuda.synchronize()
            return lg1, kv, marker
        full_lg, full_kv, _ = run(None)
        rows = [r for r in range(batch) if r % 3 != 1]
        live = torch.tensor([len(rows)] + rows + [0] * (batch - len(rows)), dtype=torch.int32, device="cuda")
        lg, kv, before = run(live)
        assert torch.equal(lg[rows], full_lg[rows])
        dead = [r for r in range(batch) if r not in rows]
        for a, b, c in zip(kv, full_kv, before):
            assert torch.equal(a[rows], b[rows])                       # live rows: the same appended codes
            assert torch.equal(a[dead], c[dead])                       # finished rows: cache untouched (position 3 still empty)
            assert bool((b[dead][:, :, :, 3] != 0).any())              # ... which the full step did write
        assert bool(torch.isfinite(lg.float()).all())
        empty = torch.zeros(1 + batch, dtype=torch.int32, device="cuda")           # nobody left: a legal call, nothing to do
        run(empty)
    finally:
        lib.wm_set_small_batch_rows(prev)


def _limits(n, seed, lo=1, hi=11):
    return torch.randint(lo, hi, (n,), generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("batch,poll", [(40, 2), (150, 3), (5, 8)])
def test_ragged_batch_equals_the_loop_that_keeps_every_row(tmpdir_module, batch, poll):
    """Rows that end at different steps (a per-row sample_len drives the raggedness: random weights never emit EOT by
    themselves; every row samples at least one token -- the reference's ranker divides by the length, W/decoding.py:104).  The loop that drops finished rows from the attention kernels and stops stepping finished groups must
    return exactly what today's loop returns (every row in every launch until the last one ends), and both must equal the
    literal reference loop cut at each row's limit.  150 utterances = three groups on three queues; group 0 of the
    40-utterance case finishes early as a whole (it is dropped at a poll point)."""
    dims = Dims(**synthetic.DIMS["micro-fullvocab"])
    eng = build_engine(tmpdir_module, "micro-fullvocab", 3)
    enc = WhisperEncoding(eng)
    mel = synthetic_mel(batch, 2 * dims.n_audio_ctx, dims.n_mels, 11).cuda()
    xa = enc.get_audio_features(mel)
    limit = _limits(batch, batch)
    if batch == 40:
        limit[:20] = _limits(20, 1, 1, 3)                   # the first group's rows all end within two tokens
    outs = []
    for skip in (True, False):
        dec = WhisperDecoding(eng)
        dec.sample_len, dec.poll_every, dec.skip_finished_rows = 14, poll, skip
        dec.detect_language(xa)
        outs.append(dec.main_loop(xa, row_limit=limit) + (dec,))
    (t_a, lp_a, nsp_a, dec_a), (t_b, lp_b, nsp_b, _) = outs
    assert torch.equal(t_a.cpu(), t_b.cpu()) and torch.equal(lp_a.cpu(), lp_b.cpu()) and nsp_a == nsp_b
    eot, L0 = dec_a.tokenizer.eot, dec_a.sample_begin
    n_sampled = ((t_a[:, L0:] != eot).sum(dim=1)).cpu()
    assert torch.equal(n_sampled, torch.minimum(limit, torch.tensor(t_a.shape[1] - L0)))        # every row ran to its own limit
    assert t_a.shape[1] == L0 + int(limit.max()) + 1                              # the loop ended where the last row did
    # against the literal reference loop (by-name decode(), host filters, no limits): identical up to each row's limit
    ref = WhisperDecoding(eng)
    ref.sample_len = int(limit.max())
    ref.tokens = dec_a.tokens.clone()
    sub = [0, batch // 2, batch - 1] + [int(i) for i in torch.argsort(limit)[-2:]]
    ref.tokens = dec_a.tokens[sub].clone()
    t_ref, lp_ref, _ = ref.main_loop_reference(xa[sub].contiguous())
    for k, r in enumerate(sub):
        n = int(limit[r])
        assert torch.equal(t_a[r, :L0 + n].cpu(), t_ref[k, :L0 + n].cpu()), r
        assert bool((t_a[r, L0 + n:] == eot).all())
    post = dec_a.post_process(t_a, lp_a, nsp_a, xa, ["en"] * batch)
    assert [len(p.tokens) for p in post] == limit.tolist()


def test_finished_rows_stop_costing_bandwidth(tmpdir_module):
    """The point of the live list, measured: a batch in which 3 of 4 rows have finished takes far less time per token in
    the cross-attention launch than the same batch with every row alive (large-v2 width, one layer; HIP events)."""
    lib = native.load_library()
    dims_d = dict(synthetic.DIMS["large-v2"], n_audio_layer=1, n_text_layer=1)
    name = "large-v2-1layer"
    synthetic.DIMS[name] = dims_d
    eng = build_engine(tmpdir_module, name, 2)
    dims = Dims(**dims_d)
    enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
    batch = 128
    mel = synthetic_mel(batch, 3000, 80, 3).cuda()
    cross = dec.xa2cross_key_value(enc.get_audio_features(mel))
    cap, H, V = dims.n_text_ctx, dims.n_text_head, dims.n_vocab
    kv = [torch.zeros((batch, 2, H, cap, 64), dtype=torch.float16, device="cuda")]
    lg = torch.zeros((batch, 1, V), dtype=torch.float16, device="cuda")
    toks = torch.full((batch, 1), 100, dtype=torch.int32, device="cuda")
    sess, pos = dec.decoder_session, dec.positional_embedding

    def timed(live):
        for _ in range(3):
            sess.decoder_step(toks, pos[0:1], cross, None, cap, kv, cap, lg, 0, stream(), live_rows=live)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            sess.decoder_step(toks, pos[0:1], cross, None, cap, kv, cap, lg, 0, stream(), live_rows=live)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 20
    rows = list(range(0, batch, 4))
    live = torch.tensor([len(rows)] + rows + [0] * (batch - len(rows)), dtype=torch.int32, device="cuda")
    t_full, t_quarter = timed(None), timed(live)
    print(f"decoder step, 1 layer, 128 utterances: {t_full * 1e3:.0f} us with every row live, {t_quarter * 1e3:.0f} us with 32 live")
    # 128 x 7.68 MB = 983 MB of K/V per step vs 246 MB: ~110 us less at the kernel's 6.5 TB/s; the rest of the step (embedding,
    # six Linears, self-attention, the 133 MB logits matrix: ~150 us) stays.  Measured: 297 -> 190 us.
    assert t_full - t_quarter > 0.6 * (0.75 * 128 * 7.68e6 / 6.6e12) * 1e3


# ------------------------------------------------------------------------------------------ sampling on the engine
def test_sampled_candidates_logprobs_match_oracle(tmpdir_module):
    """temperature 0.7, best_of 3 on the engine (device generator: the draws themselves cannot be pinned across devices --
    the host rules are pinned on CPU in tests/test_sampling_cpu.py against the reference's goldens).  What CAN be held
    to the oracle: teacher-forcing it with the candidates the engine sampled, the running log-probability of every
    candidate under Whisper's rules (W/decoding.py:287-290) equals the engine's sum_logprobs, the first sampled token is
    an allowed initial timestamp, candidates of one utterance share its audio, and the ranker picks the arg-max."""
    dims = Dims(**synthetic.DIMS["micro-fullvocab"])
    sd = synthetic_state_dict(dims, 3)
    eng = build_engine(tmpdir_module, "micro-fullvocab", 3)
    enc = WhisperEncoding(eng)
    n_audio, n_group, n_steps = 4, 3, 6
    mel = synthetic_mel(n_audio, 2 * dims.n_audio_ctx, dims.n_mels, 77)
    xa = enc.get_audio_features(mel.cuda())
    torch.manual_seed(0)
    samp = WhisperDecoding(eng, options=DecodingOptions(temperature=0.7, best_of=n_group, sample_len=n_steps))
    samp.detect_language(xa)
    toks, lps, nsp = samp.main_loop(xa)
    assert toks.shape[0] == n_audio * n_group
    toks_h, lps_h = toks.cpu().numpy(), lps.cpu().numpy()
    tk = samp.tokenizer
    L0 = samp.sample_begin
    assert len({tuple(r) for r in toks_h.tolist()}) > n_audio                  # candidates of an utterance differ
    rules = DR.RuleSet(DR.MULTILINGUAL, L0, list(samp._get_suppress_tokens()), list(tk.blank_tokens()) + [tk.eot],
                       samp.max_initial_timestamp_index)
    oracle = OracleModel(dims, sd, OracleConfig(act="float16"))
    ckv = oracle.cross_kv(oracle.encoder(mel.repeat_interleave(n_group, dim=0)))
    kv, want = None, np.zeros(n_audio * n_group, dtype=np.float64)
    alive = np.ones(n_audio * n_group, dtype=bool)
    for i in range(toks_h.shape[1] - L0):
        feed = toks_h[:, :L0] if i == 0 else toks_h[:, L0 + i - 1:L0 + i]
        logits, kv = oracle.decoder(torch.from_numpy(feed), ckv, kv)
        lg = DR.apply_filters(logits[:, -1].numpy(), toks_h[:, :L0 + i], rules)
        lp = DR.log_softmax_f32(lg)
        nxt = toks_h[:, L0 + i]
        assert np.isfinite(lp[np.arange(len(nxt)), nxt][alive]).all()         # every sampled token was allowed by the rules
        want += np.where(alive, lp[np.arange(len(nxt)), nxt], 0.0)
        alive &= nxt != tk.eot
    assert np.abs(want - lps_h).max() < n_steps * 2 * LOGIT_TOL, np.abs(want - lps_h).max()
    tb = tk.timestamp_begin
    assert ((toks_h[:, L0] >= tb) & (toks_h[:, L0] <= tb + samp.max_initial_timestamp_index)).all()
    out = samp.post_process(toks, lps, nsp, xa, ["en"] * n_audio)
    for a, r in enumerate(out):
        grp = toks_h[a * n_group:(a + 1) * n_group]
        lens = [int((g[L0:] != tk.eot).sum()) for g in grp]
        pick = int(np.argmax([lps_h[a * n_group + c] / max(lens[c], 1) for c in range(n_group)]))
        assert r.tokens == [int(t) for t in grp[pick][L0:L0 + lens[pick]]]


# ------------------------------------------------------------------------------------------ int8 KV codes, per element
def _ulp16(x):
    """Spacing of fp16 numbers at |x| (subnormal spacing below 2^-14)."""
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -14)))
    return 2.0 ** (e - 10)


@pytest.mark.parametrize("weight_only", [False, True])
def test_int8_kv_codes_differ_only_at_rounding_boundaries(tmpdir_module, weight_only):
    """Integer work is held bit-exact.  The int8 KV cache is the one integer tensor computed FROM fp16 arithmetic: code =
    sat_s8(rne(x / t)) of a k / v value x that the engine and the oracle both compute in fp16 with fp32 sums, in different
    summation orders.  So the codes must be EQUAL except where the oracle's own x sits within fp16 rounding noise of a
    code boundary (n + 1/2) t -- and that is checked element by element, with no percentage budget: for every code of the
    first layer (its inputs are exact: embedding + LayerNorm + one Linear) that differs, |difference| is one code and the
    oracle's pre-quantisation value lies within two fp16 ulps of the boundary between the two codes.
    With weight-only int8 Linears the k / v values themselves carry the weight-only arithmetic's freedom (the engine
    scales the fp32 sum of a . q once per channel, the oracle multiplies by per-element dequantised weights fp16(q . s) --
    both inside the reference's 1.5 colmax / 128 tolerance, R/tests/quantization/_utils.py:66-88): the values agree to a
    few fp16 ulps instead of one, and the proof uses 6 ulps there (measured worst case: 3.8)."""
    dims = Dims(**synthetic.DIMS["micro"])
    sd = synthetic_state_dict(dims, 7)
    mel = synthetic_mel(8, 2 * dims.n_audio_ctx, dims.n_mels, 1234)
    scales = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=weight_only)).calibrate_kv_scales(mel, 6)
    oracle = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=weight_only, int8_kv=True, kv_scales=scales))
    oracle.keep_pre_quant = {}
    n_steps = 10
    ref = greedy_reference_run(oracle, mel, [5, 17, 900], n_steps)
    eng = build_engine(tmpdir_module, "micro", 7, weight_only, True, scales)
    enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
    cross = dec.xa2cross_key_value(enc.get_audio_features(mel.cuda()))
    logits, kv = dec.decode(torch.tensor([[5, 17, 900]] * 8).cuda(), cross)
    for s in range(n_steps - 1):
        logits, kv = dec.decode(ref["ids"][:, s:s + 1].cuda(), cross, kv)
    t = float(np.float32(scales[0]))
    pre = torch.cat(oracle.keep_pre_quant[0], dim=3).numpy().astype(np.float64)          # [B,2,H,T,64]: what layer 0's codes are rounded from
    want = ref["self_kv"][0].numpy().astype(np.int64)
    assert np.array_equal(kv_quantize(torch.from_numpy(pre).float(), scales[0]).numpy(), want)
    got = kv[0].cpu().numpy().astype(np.int64)
    assert got.shape == want.shape
    diff = got - want
    idx = np.nonzero(diff)
    print(f"int8 KV, layer 0, weight_only={weight_only}: {len(idx[0])} of {diff.size} codes differ")
    assert np.abs(diff).max() <= 1
    x = pre[idx]
    boundary = (np.minimum(got[idx], want[idx]) + 0.5) * t                    # the boundary between the two codes, in value units
    dist = np.abs(x - boundary)
    assert (dist <= (6 if weight_only else 2) * _ulp16(x) + 1e-12).all(), (dist / _ulp16(x)).max()
    # deeper layers see inputs that already differ by fp16 noise amplified through a block: still never more than one code
    for layer in range(1, dims.n_text_layer):
        assert (kv[layer].cpu().int() - ref["self_kv"][layer].int()).abs().max() <= 1


# ------------------------------------------------------------------------------------------ full depth, fp16 and W8
@pytest.mark.parametrize("weight_only", [False, True])
def test_full_depth_large_v2_fp16_and_weight_only_match_oracle_on_gpu(tmpdir_module, weight_only):
    """BASELINE.json configs[1] and [2] (large-v2 fp16; weight-only int8) at FULL depth -- 32 + 32 layers -- against the
    oracle with its parameters on the GPU, teacher-forced, like test_full_size_large_v2_matches_oracle_on_gpu does for
    configs[3]: the small-model tolerance holds at depth 32."""
    dims = Dims(**synthetic.DIMS["large-v2"])
    ck = synthetic.synthetic_checkpoint("large-v2", 9, device="cuda")
    sd = {k: v.cpu() for k, v in ck["model_state_dict"].items()}
    mel = synthetic_mel(2, 3000, 80, 4242).cuda()
    oracle = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=weight_only)).to("cuda")
    del sd
    prompt = [dims.n_vocab - 1607, dims.n_vocab - 1606, dims.n_vocab - 1506]
    n_steps = 4
    ref = greedy_reference_run(oracle, mel, prompt, n_steps)
    del oracle
    out = os.path.join(tmpdir_module, f"eng_large-v2_full_{int(weight_only)}")
    argv = ["--output_dir", out, "--use_gpt_attention_plugin", "--use_gemm_plugin", "--use_layernorm_plugin", "--log_level", "error"]
    if weight_only:
        argv.append("--use_weight_only")
    B.build_from_checkpoint(ck, B.parse_arguments(argv))
    del ck
    torch.cuda.empty_cache()
    enc, dec = WhisperEncoding(Path(out)), WhisperDecoding(Path(out))
    xa = enc.get_audio_features(mel)
    d_xa = float((xa.float() - ref["xa"]).abs().max())
    cross = dec.xa2cross_key_value(xa)
    d_ckv = max(float((c.float() - r).abs().max()) for c, r in zip(cross, ref["cross_kv"]))
    logits, kv = dec.decode(torch.tensor([prompt] * 2).cuda(), cross)
    worst = float((logits.float() - ref["logits"][0]).abs().max())
    n_safe = n_ok = 0
    for s_ in range(n_steps - 1):
        logits, kv = dec.decode(ref["ids"][:, s_:s_ + 1], cross, kv)
        worst = max(worst, float((logits[:, 0].float() - ref["logits"][s_ + 1][:, 0]).abs().max()))
        safe = (ref["margins"][:, s_ + 1] > 2 * LOGIT_TOL).cpu().numpy()
        got = logits[:, 0].float().argmax(-1).cpu().numpy()
        n_safe += int(safe.sum())
        n_ok += int((got[safe] == ref["ids"][:, s_ + 1].cpu().numpy()[safe]).sum())
    print(f"full depth, weight_only={weight_only}: max|xa| = {d_xa:.4f}, max|cross K/V| = {d_ckv:.4f}, max|logits| = {worst:.4f}, ids {n_ok}/{n_safe}")
    assert d_xa < 5e-2 and d_ckv < 5e-2 and worst < LOGIT_TOL, (d_xa, d_ckv, worst)
    assert n_ok == n_safe and n_safe > 0
    assert kv[0].dtype == torch.float16


# ------------------------------------------------------------------------------------------ bounds
def test_cross_attention_split_bound(lib):
    """The merge kernel of the key-split cross-attention combines at most 16 partial results: 16 splits are exact against
    the oracle, 17 are refused (before this check they were silently dropped)."""
    r = np.random.Generator(np.random.Philox(16))
    B_, L, H, Tk = 2, 1, 3, 1500
    C_ = H * 64
    q = r.standard_normal((B_ * L, C_)).astype(np.float16).astype(np.float32)
    kv = r.standard_normal((B_, 2, H, Tk, 64)).astype(np.float16)
    m = OracleModel(Dims(80, Tk, C_, H, 0, 8, 4, C_, H, 0), {}, OracleConfig(act="float16"))
    kk = torch.from_numpy(kv[:, 0]).float().permute(0, 2, 1, 3).reshape(B_, Tk, C_)
    vv = torch.from_numpy(kv[:, 1]).float().permute(0, 2, 1, 3).reshape(B_, Tk, C_)
    ref = m._attend(torch.from_numpy(q).reshape(B_, L, C_), kk, vv, H).numpy().reshape(B_ * L, C_)
    qd, kvd = torch.from_numpy(q).cuda(), torch.from_numpy(kv).cuda()
    out = torch.zeros((B_ * L, C_), dtype=torch.float16, device="cuda")
    for nsplit in (16, 9):
        ws = torch.zeros(B_ * H * nsplit * L * 66, dtype=torch.float32, device="cuda")
        native.check(lib.wm_attn_decode_cross(qd.data_ptr(), B_, L, H, Tk, kvd.data_ptr(), out.data_ptr(), nsplit, ws.data_ptr(), stream()))
        torch.cuda.synchronize()
        assert np.abs(out.float().cpu().numpy() - ref).max() <= 2e-3, nsplit
    ws = torch.zeros(B_ * H * 17 * L * 66, dtype=torch.float32, device="cuda")
    with pytest.raises(native.WmError, match="nsplit=17"):
        native.check(lib.wm_attn_decode_cross(qd.data_ptr(), B_, L, H, Tk, kvd.data_ptr(), out.data_ptr(), 17, ws.data_ptr(), stream()))


def test_fused_linear_refuses_widths_it_would_overrun(lib):
    """wm_gemv_fused writes whole 16-column blocks in modes 0-2: leading dimensions narrower than 16 * n_blocks, or a
    logical width that is not the blocks' width, are errors (they used to be out-of-bounds writes)."""
    a = torch.zeros((2, 64), dtype=torch.float16, device="cuda")
    wt = torch.zeros((4, 2, 64, 8), dtype=torch.float16, device="cuda")          # 4 blocks = 64 columns, K = 64
    bias = torch.zeros(64, dtype=torch.float16, device="cuda")
    o32 = torch.zeros((2, 64), dtype=torch.float32, device="cuda")
    o16 = torch.zeros((2, 64), dtype=torch.float16, device="cuda")
    x = torch.zeros((2, 64), dtype=torch.float16, device="cuda")

    def call(**kw):
        io = native.WmGemvIO()
        io.a, io.lda, io.m, io.k = a.data_ptr(), 64, 2, 64
        io.wt, io.n_blocks, io.w8 = wt.data_ptr(), 4, 0
        io.bias, io.gelu_kind = bias.data_ptr(), 1
        io.out32, io.ld32, io.out16, io.ld16, io.n_valid, io.x, io.ldx = o32.data_ptr(), 64, o16.data_ptr(), 64, 64, x.data_ptr(), 64
        for k, v in kw.items():
            setattr(io, k, v)
        native.check(lib.wm_gemv_fused(C.byref(io), stream()))
    for mode in (0, 1, 2):
        call(mode=mode)
    torch.cuda.synchronize()
    with pytest.raises(native.WmError, match="ld32"):
        call(mode=0, ld32=48)
    with pytest.raises(native.WmError, match="ld16"):
        call(mode=1, ld16=60)
    with pytest.raises(native.WmError, match="ldx"):
        call(mode=2, ldx=32)
    with pytest.raises(native.WmError, match="n_valid"):
        call(mode=1, n_valid=50)
    call(mode=3, n_valid=50)                                            # the logits mode is the one with a ragged last block
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------ data-parallel readiness
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_line(extra, cache, timeout=1500):
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--model", "large-v2-6layer", "--config", "int8", "--decode-steps", "24",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--length-dist", "forced", "--encoder-cus", "0",
           "--engine-cache", cache] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_rccl_streams_do_not_cost_the_decode_loop_its_queues(tmp_path):
    """The decode loop's utterance groups each own a hardware queue (streams created with a full CU mask are never
    multiplexed).  RCCL creates streams of its own; DESIGN.md section 5 records 23.5 instead of 18.1 ms per step when a stray
    stream made two groups share a queue.  The same three-group job with torch.distributed over RCCL initialised
    (--force-dist: scatter and gather run as collectives) must step as fast as the plain one."""
    cache = str(tmp_path / "engines")
    plain = _bench_line(["--gpus", "1", "--batch", "192"], cache)
    rccl = _bench_line(["--gpus", "1", "--batch", "192", "--force-dist"], cache)
    print(f"ms per step: plain {plain['ms_per_step']}, with RCCL initialised {rccl['ms_per_step']}")
    assert rccl["n_gpus"] == 1 and rccl["ms_per_step"] < 1.08 * plain["ms_per_step"]


def test_two_gpus_scale_like_one(tmp_path):
    """`bench.py --gpus 2` (it starts its two ranks itself: one process per GPU, utterances scattered / token ids gathered
    over RCCL, nothing reduced) on boxes that have two GPUs: the per-GPU rate must stay within 5 % of the one-GPU run --
    weak scaling of a path with no data-path collective.  Skipped on one-GPU boxes (the pool this suite runs on)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    cache = str(tmp_path / "engines")
    one = _bench_line(["--gpus", "1", "--batch", "192"], cache)
    two = _bench_line(["--gpus", "2", "--batch", "192"], cache)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["value"] / 2 > 0.95 * one["value"], (one["value"], two["value"])


# ------------------------------------------------------------------------- round 3, third session: small-batch chain
@pytest.mark.parametrize("batch", [1, 6])
def test_self_attention_wave_forms_agree_in_the_engine(tmpdir_module, lib, batch):
    """The one-wave and the four-wave form of the decode self-attention inside the engine: the appended cache codes are
    identical, the logits agree to the fp32 summation order (a few fp16 ulps at most on the micro model)."""
    dims = Dims(**synthetic.DIMS["micro"])
    eng = build_engine(tmpdir_module, "micro", 7, True, True, [0.05, 0.06])
    enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
    mel = synthetic_mel(batch, 2 * dims.n_audio_ctx, dims.n_mels, 11).cuda()
    cross = dec.xa2cross_key_value(enc.get_audio_features(mel))
    cap, H, V = dims.n_text_ctx, dims.n_text_head, dims.n_vocab
    sess, pos = dec.decoder_session, dec.positional_embedding
    n_steps = min(cap - 4, 12)
    toks = torch.randint(0, V, (batch, 3 + n_steps), generator=torch.Generator().manual_seed(3)).to(torch.int32).cuda()

    def run(waves):
        prev = lib.wm_set_self_attn_waves(waves)
        try:
            kv = [torch.zeros((batch, 2, H, cap, 64), dtype=torch.int8, device="cuda") for _ in range(dims.n_text_layer)]
            lg0 = torch.zeros((batch, 3, V), dtype=torch.float16, device="cuda")
            sess.decoder_step(toks[:, :3].contiguous(), pos[0:3], cross, None, cap, kv, cap, lg0, 0, stream())
            out = [lg0[:, -1]]
            for t in range(n_steps):
                lg = torch.zeros((batch, 1, V), dtype=torch.float16, device="cuda")
                sess.decoder_step(toks[:, 3 + t:4 + t].contiguous(), pos[3 + t:4 + t], cross, kv, cap, kv, cap, lg, 3 + t, stream())
                out.append(lg[:, 0])
            torch.cuda.synchronize()
            return torch.stack(out, 1).float(), kv
        finally:
            lib.wm_set_self_attn_waves(prev)
    l1, kv1 = run(1)
    l4, kv4 = run(4)
    assert float((l1 - l4).abs().max()) <= LOGIT_TOL_INT8_KV
    same = sum(int((a == b).sum()) for a, b in zip(kv1, kv4)); total = sum(a.numel() for a in kv1)
    assert same >= 0.99 * total           # (the codes of layer > 0 inherit the attention's last-bit differences through the residual stream)
    assert torch.equal(kv1[0], kv4[0])    # layer 0's k / v are functions of the token and its position alone
