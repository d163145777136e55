"""Round 6 GPU tests (through the C ABI, on a real MI355X): stream-parallel utterance groups never take a one-launch decode step, a give-up
found by somebody else is said loudly, the by-name decode() looks at the give-up word itself."""
import ctypes as C
import logging
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

import native  # noqa: E402
import synthetic  # noqa: E402
from decoding import DecodingOptions, WhisperDecoding  # noqa: E402
from encoding import WhisperEncoding  # noqa: E402
from oracle.whisper_oracle import Dims, synthetic_mel  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return native.load_library()


@pytest.fixture()
def chain_rearmed(lib):
    """Every test here starts with the one-launch forms armed at the default mode and leaves them so."""
    lib.wm_set_decode_chain(-1)
    err = C.c_int(0)
    native.check(lib.wm_decode_chain_error(C.byref(err)))
    lib.wm_set_decode_chain(-1)
    yield
    native.check(lib.wm_decode_chain_error(C.byref(err)))
    lib.wm_set_decode_chain(-1)


def _engine(tmp_path_factory, name):
    from test_gpu_model import build_engine
    model = "large-v2-6layer"
    dims = Dims(**synthetic.DIMS[model])
    scales = [0.05 + 0.01 * i for i in range(dims.n_text_layer)]
    return build_engine(tmp_path_factory.mktemp(name), model, 3, True, True, scales), dims


@pytest.mark.parametrize("n_batch,micro", [(16, None), (8, 2), (12, 3)])
def test_stream_parallel_groups_of_up_to_eight_rows_never_take_the_one_launch_step(lib, tmp_path_factory, chain_rearmed, n_batch, micro):
    """`_groups` cuts 16 utterances into 2 x 8 (and `micro_batches` = 2 / 3 cuts 8 / 12 into 2 x 4 / 3 x 4): every group is within
    the one-launch step's eight rows, and two such 256-workgroup launches replayed side by side on two streams each hold half of the chip and
    wait for the other half until the bounded waits give up (profiles/r6a_b16_default.err: the B = 16 line of round 5 was a give-up, a
    re-decode and a device taken off the form).  The loop now says `not_alone` (ABI 8) for every step of a stream-parallel group: the
    library takes a launch per kernel.  After a whole loop, eager and replayed: no chain launch was issued, nothing declined, no error
    pending -- and the tokens, log-probabilities and caches are those of the launch-per-kernel mode (wm_set_decode_chain(0))."""
    eng, dims = _engine(tmp_path_factory, f"par{n_batch}")
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(n_batch, 2 * dims.n_audio_ctx, dims.n_mels, 91).cuda())
    outs = []
    for on in (0, 2):
        lib.wm_set_decode_chain(on)
        dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=10))
        dec.micro_batches = micro
        n_micro, bounds = dec._groups(n_batch)
        assert n_micro > 1 and all(hi - lo <= 8 for lo, hi in bounds), (n_micro, bounds)
        dec.detect_language(xa)
        for use_graphs in (False, True):
            dec.use_graphs = use_graphs
            for st in dec._state.values():
                st['graphs'].clear()
            before = native.chain_status()
            t, lp, _ = dec.main_loop(xa, ignore_eot=True)
            after = native.chain_status()
            assert after["launches"] == before["launches"], (on, use_graphs, "a stream-parallel group took a one-launch step")
            assert not after["error_pending"] and not after["declined"], after
            outs.append((on, use_graphs, t.cpu(), lp.cpu(), [c.clone() for c in dec._state[n_batch]['kv']]))
        del dec
    ref = outs[0]
    for on, use_graphs, t, lp, kv in outs[1:]:
        assert torch.equal(t, ref[2]) and torch.equal(lp, ref[3]), (on, use_graphs)
        for a, b in zip(kv, ref[4]):
            assert torch.equal(a, b), (on, use_graphs)


def test_not_alone_declines_the_one_launch_step_at_the_c_abi(lib, tmp_path_factory, chain_rearmed):
    """The same promise one level down: wm_decoder_step with `not_alone` set issues no chain launch for a group the form would serve, and
    produces the same logits and cache bytes as the call without it (which does take the form)."""
    eng, dims = _engine(tmp_path_factory, "notalone")
    enc = WhisperEncoding(eng)
    rows = 4
    xa = enc.get_audio_features(synthetic_mel(rows, 2 * dims.n_audio_ctx, dims.n_mels, 92).cuda())
    lib.wm_set_decode_chain(2)
    dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=4))
    st = dec._fast_state(rows, xa.device)
    cross = dec._cross_persistent(xa, st)
    cap, V = dec.decoder_config['num_text_ctx'], dec.decoder_config['vocab_size']
    tok = torch.full((rows, 1), dec.tokenizer.sot, dtype=torch.int32, device=xa.device)
    res = []
    for not_alone in (False, True):
        for t in st['kv']:
            t.zero_()
        logits = torch.empty((rows, 1, V), dtype=torch.float16, device=xa.device)
        before = native.chain_status()["launches"]
        dec.decoder_session.decoder_step(tok, dec.positional_embedding[0:1], cross, None, cap, st['kv'], cap, logits, 0,
                                         torch.cuda.current_stream().cuda_stream, not_alone=not_alone)
        torch.cuda.synchronize()
        took = native.chain_status()["launches"] > before
        assert took == (not not_alone), (not_alone, took)
        res.append((logits.clone(), [t.clone() for t in st['kv']]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    assert not native.chain_status()["error_pending"]


def test_a_give_up_found_by_somebody_else_is_said_at_error_level_and_drops_every_instances_graphs(lib, tmp_path_factory, chain_rearmed, caplog):
    """main_loop used to acknowledge a pending give-up it did not cause with a one-time warning -- the earlier caller
    never learnt that its results were invalid.  A give-up is provoked under instance A's feet through the C ABI (wm_debug_occupy holds
    half of the CUs' LDS while a one-launch step of a bare wm_decoder_step is dispatched), nobody looks; instance B's main_loop then finds
    the word: an ERROR record naming the earlier results as invalid, B's own result correct (the launch-per-kernel path), and the captured
    graphs of instance A -- which replay chain launches -- gone as well."""
    eng, dims = _engine(tmp_path_factory, "foreign")
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(1, 2 * dims.n_audio_ctx, dims.n_mels, 93).cuda())
    lib.wm_set_decode_chain(0)
    ref_dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=8))
    ref_dec.detect_language(xa)
    t_ref, lp_ref, _ = ref_dec.main_loop(xa, ignore_eot=True)
    lib.wm_set_decode_chain(2)
    a = WhisperDecoding(eng, options=DecodingOptions(sample_len=8))
    a.detect_language(xa)
    t_a, _, _ = a.main_loop(xa, ignore_eot=True)                   # captures graphs with chain launches
    assert torch.equal(t_a.cpu(), t_ref.cpu())
    assert any(st['graphs'] for st in a._state.values())
    # a one-launch step through the bare C ABI while half of the chip cannot take its workgroups: it gives up, and this caller never looks
    st = a._fast_state(1, xa.device)
    side = torch.cuda.Stream()
    native.check(lib.wm_debug_occupy(128, 100 * 1024, 3_000_000, side.cuda_stream), "wm_debug_occupy")
    time.sleep(0.05)                 # (the occupying workgroups are resident before the one-launch step is dispatched)
    cap, V = a.decoder_config['num_text_ctx'], a.decoder_config['vocab_size']
    tok = torch.full((1, 1), a.tokenizer.sot, dtype=torch.int32, device=xa.device)
    logits = torch.empty((1, 1, V), dtype=torch.float16, device=xa.device)
    a.decoder_session.decoder_step(tok, a.positional_embedding[0:1], st['cross'], None, cap, st['kv'], cap, logits, 0,
                                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    if not native.chain_status()["error_pending"]:
        pytest.skip("the occupied CUs did not keep the one-launch step from running on this box")
    b = WhisperDecoding(eng, options=DecodingOptions(sample_len=8))
    with caplog.at_level(logging.ERROR, logger="whisper_mi355"):
        b.detect_language(xa)
        t_b, lp_b, _ = b.main_loop(xa, ignore_eot=True)
    assert any("INVALID" in r.getMessage() and r.levelno >= logging.ERROR for r in caplog.records), [r.getMessage() for r in caplog.records]
    assert torch.equal(t_b.cpu(), t_ref.cpu()) and torch.equal(lp_b.cpu(), lp_ref.cpu())
    assert not any(st['graphs'] for st in a._state.values()), "instance A still holds graphs that replay chain launches"
    stt = native.chain_status()
    assert stt["declined"] and not stt["error_pending"]


def test_decode_by_name_looks_at_the_give_up_word_itself(lib, tmp_path_factory, chain_rearmed):
    """the by-name decode() of one token for up to eight utterances with an empty cache (the reference's language pass,
    W/decoding.py:703-741) qualifies for the one-launch step and never looked at the word.  It synchronises anyway, so it looks now: with
    half of the chip occupied the call gives up, acknowledges, runs again with a launch per kernel and returns the right logits."""
    eng, dims = _engine(tmp_path_factory, "byname")
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(2, 2 * dims.n_audio_ctx, dims.n_mels, 94).cuda())
    lib.wm_set_decode_chain(0)
    dec = WhisperDecoding(eng)
    cross = dec.xa2cross_key_value(xa)
    x = torch.tensor([[dec.tokenizer.sot]] * 2).cuda()
    ref, _ = dec.decode(x, cross)
    lib.wm_set_decode_chain(2)
    before = native.chain_status()["launches"]
    got, _ = dec.decode(x, cross)
    assert native.chain_status()["launches"] > before, "the by-name language pass no longer takes the one-launch step: the test is vacuous"
    assert torch.equal(got, ref)
    side = torch.cuda.Stream()
    native.check(lib.wm_debug_occupy(128, 100 * 1024, 3_000_000, side.cuda_stream), "wm_debug_occupy")
    time.sleep(0.05)                 # (the occupying workgroups are resident before the one-launch step is dispatched)
    got2, _ = dec.decode(x, cross)
    torch.cuda.synchronize()
    assert torch.equal(got2, ref)
    assert not native.chain_status()["error_pending"]


# ------------------------------------------------------------------------------------------ the M = 1 arithmetic (VERDICT r5 "missing" 3)
@pytest.mark.parametrize("name,K,N", [("qkv", 1280, 3840), ("out", 1280, 1280), ("cross q", 1280, 1280), ("cross out", 1280, 1280),
                                       ("mlp1", 1280, 5120), ("mlp2", 5120, 1280)])
def test_one_row_linear_is_inside_the_reference_tolerance_of_both_of_its_contracts(lib, name, K, N):
    """The reference's plugin computes a one-row weight-only Linear with a GEMV kernel that rounds every product to fp16
    (weightOnlyMatrixVectorMultiplication.cu:187) and a Linear of more rows with CUTLASS (exact products): two contracts.  The engine's
    one-row Linear (wm_gemv_fused = gemv_small.hip, the kernel behind batch 1; exact fp16 x fp16 products on the matrix cores, fp32 sums,
    the per-channel scale applied to the fp32 sum) on the six shapes of a large-v2 decoder layer: inside the reference's own tolerance
    (1.5 * max / 128, R/tests/quantization/_utils.py:66-88) of the GEMV kernel's restatement (oracle.woq_gemv_reference), of the CUTLASS
    contract (the oracle's Linear) and of the reference test's ground truth -- by a factor of ten or more."""
    import numpy as np
    import weight as W
    from oracle.whisper_oracle import dequantize_int8, symmetric_quantize_int8, woq_colwise_atol, woq_gemv_reference, woq_reference_matmul
    r = np.random.default_rng(K * 7 + N)
    w = (r.standard_normal((N, K)) * 2 / np.sqrt(K)).astype(np.float16)
    x = r.standard_normal((1, K)).astype(np.float16)
    q, s = symmetric_quantize_int8(w)
    tiles = W.tile_linear(q)
    t_dev = torch.from_numpy(tiles.view(np.uint8)).cuda()
    s_dev, a_dev = torch.from_numpy(s).cuda(), torch.from_numpy(x).cuda()
    out32 = torch.full((1, N), float("nan"), dtype=torch.float32, device="cuda")
    io = native.WmGemvIO()
    io.a, io.lda, io.m, io.k = a_dev.data_ptr(), K, 1, K
    io.wt, io.n_blocks, io.w8 = t_dev.data_ptr(), N // 16, 1
    io.scale, io.mode, io.gelu_kind = s_dev.data_ptr(), 0, 1
    io.out32, io.ld32 = out32.data_ptr(), N
    native.check(lib.wm_gemv_fused(C.byref(io), torch.cuda.current_stream().cuda_stream), "wm_gemv_fused")
    torch.cuda.synchronize()
    got = out32.cpu().numpy().astype(np.float16).astype(np.float32)                        # the Linear's fp16 output
    gemv = woq_gemv_reference(x, q.T, s).astype(np.float32)
    cutlass = (x.astype(np.float32) @ dequantize_int8(q, s).astype(np.float32).T).astype(np.float16).astype(np.float32)
    truth = woq_reference_matmul(x, q.T, s).astype(np.float32)
    atol = float(woq_colwise_atol(truth)[0])
    d_gemv, d_cut, d_truth = (float(np.abs(got - y).max()) for y in (gemv, cutlass, truth))
    print(f"{name}: |engine - GEMV kernel| {d_gemv:.4g}, |engine - CUTLASS| {d_cut:.4g}, |engine - (x @ q) * s| {d_truth:.4g}, tolerance {atol:.4g}")
    assert d_gemv <= atol / 10 and d_cut <= atol / 10 and d_truth <= atol / 10
    assert d_truth <= d_gemv + 1e-6 or d_cut <= d_gemv + 1e-6      # the engine sits on the exact-product side of the two


# ------------------------------------------------------------------------------------------ language pass + prefill from graphs
@pytest.mark.parametrize("n_batch", [1, 4, 12, 16])
def test_language_pass_and_prefill_replayed_from_graphs_are_bit_identical(lib, tmp_path_factory, chain_rearmed, n_batch):
    """Round 6: from the second batch on, the language pass (where it is a launch per kernel: more than eight rows, or groups side by side) and
    the prefill + first greedy step are REPLAYED from graphs captured on the first batch -- issued eagerly their ~ 290 + 390 launches per group
    are host-bound at small batches.  Three batches of different audio through a decoder with the replay and one without: languages, language
    probabilities, tokens, log-probabilities, no-speech probabilities and every byte of the cache are identical batch by batch; the replaying
    decoder holds the graphs it should (and none for a pass that is the one-launch step)."""
    eng, dims = _engine(tmp_path_factory, f"pre{n_batch}")
    enc = WhisperEncoding(eng)
    xas = [enc.get_audio_features(synthetic_mel(n_batch, 2 * dims.n_audio_ctx, dims.n_mels, 70 + k).cuda()) for k in range(3)]
    res = {}
    for replay in (False, True):
        dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=6))
        dec.graph_prefill = replay
        out = []
        for xa in xas:
            langs, probs = dec.detect_language(xa)
            t, lp, nsp = dec.main_loop(xa, ignore_eot=True)
            out.append((langs, probs, t.cpu(), lp.cpu(), list(nsp), [c.clone() for c in dec._state[n_batch]['kv']]))
        keys = [k for st in dec._state.values() for k in st['graphs'] if isinstance(k, tuple) and ('prefill' in k or 'lang' in k)]
        res[replay] = (out, keys)
        del dec
    assert res[False][1] == []
    kinds = {("prefill" if "prefill" in k else "lang") for k in res[True][1]}
    assert "prefill" in kinds
    assert ("lang" in kinds) == (n_batch > 8), res[True][1]        # up to eight rows alone: the language pass is the one-launch step, nothing to capture
    for a, b in zip(res[False][0], res[True][0]):
        assert a[0] == b[0] and a[1] == b[1]
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and a[4] == b[4]
        for x, y in zip(a[5], b[5]):
            assert torch.equal(x, y)
    assert not torch.equal(res[True][0][0][2], res[True][0][1][2])     # (different audio: different tokens -- the replay reads the new batch's data)
    st_ = native.chain_status()
    assert not st_["error_pending"] and not st_["declined"]


def test_small_batches_are_not_encoded_beside_the_decode_loop(lib, tmp_path_factory, chain_rearmed):
    """Round 6: WhisperEncoding.prefetch() of a batch of up to eight clips does not start a pass beside the decode loop (whose token step is
    the one-launch form: it wants every CU, and a step dispatched within ~ 2 ms of the start of a budget-confined pass gave up in 5 of 25
    bench runs at five utterances); collect() runs the pass.  Same audio features bit for bit, no helper thread, and five ragged batches
    through the pipelined schedule of bench.py's second figure leave the one-launch step armed."""
    import threading
    eng, dims = _engine(tmp_path_factory, "defer")
    enc = WhisperEncoding(eng)
    mel = synthetic_mel(5, 2 * dims.n_audio_ctx, dims.n_mels, 61).cuda()
    ref = enc.get_audio_features(mel).clone()
    n_threads = threading.active_count()
    enc.prefetch(mel, 96)
    assert threading.active_count() == n_threads and enc._prefetch[0] is None
    enc.loop_ended()
    got = enc.collect()
    torch.cuda.synchronize()
    assert torch.equal(got, ref) and enc._prefetch is None and enc.last_release_layer == 0
    big = synthetic_mel(9, 2 * dims.n_audio_ctx, dims.n_mels, 62).cuda()
    enc.prefetch(big, 96)
    assert enc._prefetch[0] is not None                      # nine clips: the helper thread, as before
    ref9 = enc.collect().clone()
    assert torch.equal(ref9, enc.get_audio_features(big))
    dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=12))
    limits = [[3, 9, 5, 12, 7], [2, 2, 11, 4, 6], [12, 1, 8, 8, 3], [5, 5, 5, 5, 5], [1, 12, 2, 10, 4]]
    xa = enc.get_audio_features(mel)
    before = native.chain_status()["launches"]
    for k, lim in enumerate(limits):
        dec.detect_language(xa)
        if k + 1 < len(limits):
            enc.prefetch(mel, 96)
        t, _, _ = dec.main_loop(xa, row_limit=torch.tensor(lim, dtype=torch.int32))
        enc.loop_ended()
        if k + 1 < len(limits):
            xa = enc.collect()
        for b in range(5):
            assert int((t[b, 3:] != 50257).sum()) == lim[b], (k, b)
    st_ = native.chain_status()
    assert st_["launches"] > before and not st_["declined"] and not st_["error_pending"], st_
