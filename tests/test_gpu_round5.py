"""Round 5 GPU tests (through the C ABI, on a real MI355X): the one-launch decode step at its product shape and under stress."""
import ctypes as C
import logging
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
from oracle.whisper_oracle import Dims, OracleConfig, OracleModel, greedy_reference_run, synthetic_mel  # noqa: E402

LOGIT_TOL = 3e-2             # tests/test_gpu_model.py: fp16 activations, fp32 accumulate
LOGIT_TOL_INT8_KV = 6e-2     # ... plus one LSB of the int8 cache


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return native.load_library()


@pytest.fixture(scope="module")
def tmpdir_module(tmp_path_factory):
    return str(tmp_path_factory.mktemp("engines5"))


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


def _write_kv_scales(qdir, scales):
    os.makedirs(qdir, exist_ok=True)
    for i, s in enumerate(scales):
        np.array([s], dtype=np.float32).tofile(os.path.join(qdir, f"model.decoder.blocks.{i}.attn.query_key_value.scale_y_quant_orig.bin"))


def _small_engine(tmp, model="large-v2-6layer", weight_only=True, int8_kv=True):
    from test_gpu_model import build_engine
    dims = Dims(**synthetic.DIMS[model])
    scales = [0.05 + 0.01 * i for i in range(dims.n_text_layer)] if int8_kv else None
    return build_engine(tmp, model, 3, weight_only, int8_kv, scales), dims


# ------------------------------------------------------------------------------------------ the default batch-1 path at FULL depth
@pytest.mark.parametrize("quantised", [True, False])
def test_full_depth_batch1_one_launch_step_matches_oracle_and_the_launch_per_kernel_path(lib, tmpdir_module, chain_rearmed, quantised):
    """The reference's operating point (W/run.py:43-61, W/decoding.py:785-821: ONE utterance) on the product's default path at the
    product's shape: large-v2, 32 + 32 layers -- exactly CHAIN_MAX_LAYERS of csrc/gemv_chain.hip -- weight-only int8 + int8 KV
    (BASELINE.json configs[3]) and fp16 (configs[1]).
      (a) batch-1 `main_loop` with the decode step as a launch per kernel / one launch per layer / ONE launch per token step, eagerly
          and replayed from the captured graph: token ids, log-probabilities and every byte of the KV cache identical;
      (b) the one-launch step teacher-forced with the GPU-resident oracle's ids through the in-place-cache call `main_loop` makes:
          logits within the tolerance of the small models at every step, and bit-identical to the launch-per-kernel path's;
      (c) the launches really were the one-launch form (wm_decode_chain_status counts them) and none gave up."""
    dims = Dims(**synthetic.DIMS["large-v2"])
    assert dims.n_text_layer == 32
    ck = synthetic.synthetic_checkpoint("large-v2", 9, device="cuda")
    sd = {k: v.cpu() for k, v in ck["model_state_dict"].items()}
    mel = synthetic_mel(1, 3000, 80, 4242).cuda()
    scales = None
    if quantised:
        cal = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=True)).to("cuda")
        scales = cal.calibrate_kv_scales(mel, 3)
        del cal
    oracle = OracleModel(dims, sd, OracleConfig(act="float16", weight_only=quantised, int8_kv=quantised, kv_scales=scales)).to("cuda")
    del sd
    prompt = [dims.n_vocab - 1607, dims.n_vocab - 1606, dims.n_vocab - 1506]
    n_steps = 6
    ref = greedy_reference_run(oracle, mel, prompt, n_steps)
    del oracle
    out = os.path.join(tmpdir_module, f"eng_large-v2_b1_{int(quantised)}")
    argv = ["--output_dir", out, "--use_gpt_attention_plugin", "--use_gemm_plugin", "--use_layernorm_plugin", "--log_level", "error"]
    if quantised:
        qdir = os.path.join(tmpdir_module, "quantize_large-v2_b1", "1-gpu")
        _write_kv_scales(qdir, scales)
        argv += ["--use_weight_only", "--int8_kv_cache", "--quantize_dir", qdir]
    B.build_from_checkpoint(ck, B.parse_arguments(argv))
    del ck
    torch.cuda.empty_cache()
    tol = LOGIT_TOL_INT8_KV if quantised else LOGIT_TOL
    enc = WhisperEncoding(Path(out))
    xa = enc.get_audio_features(mel)

    # ---- (b) teacher-forced, through Session.decoder_step with the in-place cache (what main_loop issues per token) ------------
    def teacher_forced(mode):
        lib.wm_set_decode_chain(mode)
        dec = WhisperDecoding(Path(out))
        cfg = dec.decoder_config
        cap, V = cfg['num_text_ctx'], cfg['vocab_size']
        st = dec._fast_state(1, xa.device)
        cross = dec._cross_persistent(xa, st)
        tokens = torch.zeros((1, cap + 1), dtype=torch.int32, device="cuda")
        tokens[0, :3] = torch.tensor(prompt, dtype=torch.int32)
        tokens[0, 3:3 + n_steps - 1] = ref["ids"][0, :n_steps - 1].to(torch.int32)
        s = torch.cuda.current_stream().cuda_stream
        sess, pos = dec.decoder_session, dec.positional_embedding
        before = native.chain_status()["launches"]
        logits3 = torch.empty((1, 3, V), dtype=torch.float16, device="cuda")
        sess.decoder_step(tokens[:, :3], pos[0:3], cross, None, cap, st['kv'], cap, logits3, 0, s)
        got = [logits3.clone()]
        logits1 = torch.empty((1, 1, V), dtype=torch.float16, device="cuda")
        for k in range(n_steps - 1):
            cur = 3 + k + 1
            sess.decoder_step(tokens[:, cur - 1:cur], pos[cur - 1:cur], cross, st['kv'], cap, st['kv'], cap, logits1, cur - 1, s)
            got.append(logits1.clone())
        torch.cuda.synchronize()
        launches = native.chain_status()["launches"] - before
        kv = [c[:, :, :, :3 + n_steps - 1].clone() for c in st['kv']]
        del dec, st
        return got, kv, launches

    got2, kv2, n2 = teacher_forced(2)
    got0, kv0, n0 = teacher_forced(0)
    assert n2 == n_steps - 1 and n0 == 0, (n2, n0)              # (c) one launch per token step | none
    worst = float((got2[0].float() - ref["logits"][0]).abs().max())
    n_safe = n_ok = 0
    for k in range(n_steps - 1):
        worst = max(worst, float((got2[k + 1][:, 0].float() - ref["logits"][k + 1][:, 0]).abs().max()))
        safe = bool(ref["margins"][0, k + 1] > 2 * tol)
        n_safe += int(safe)
        n_ok += int(safe and int(got2[k + 1][0, 0].float().argmax()) == int(ref["ids"][0, k + 1]))
    print(f"full depth, batch 1, one launch per token step, quantised={quantised}: max|logits - oracle| = {worst:.4f}, ids {n_ok}/{n_safe}")
    assert worst < tol, worst
    assert n_ok == n_safe and n_safe > 0
    for a, b in zip(got2, got0):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    for a, b in zip(kv2, kv0):
        assert torch.equal(a, b)

    # ---- (a) the loop itself, three forms x eager / replayed --------------------------------------------------------------------
    outs = []
    for mode in (0, 1, 2):
        lib.wm_set_decode_chain(mode)
        dec = WhisperDecoding(Path(out), options=DecodingOptions(sample_len=10))
        dec.detect_language(xa)
        for use_graphs in (False, True):
            dec.use_graphs = use_graphs
            for st in dec._state.values():
                st['graphs'].clear()
            before = native.chain_status()["launches"]
            t, lp, _ = dec.main_loop(xa, ignore_eot=True)
            n_l = native.chain_status()["launches"] - before
            assert (n_l > 0) == (mode > 0), (mode, use_graphs, n_l)
            outs.append((mode, use_graphs, t.cpu(), lp.cpu(), [c.clone() for c in dec._state[1]['kv']]))
        del dec
    st_ = native.chain_status()
    assert not st_["error_pending"] and not st_["declined"], st_
    r = outs[0]
    for mode, use_graphs, t, lp, kv in outs[1:]:
        assert torch.equal(t, r[2]) and torch.equal(lp, r[3]), (mode, use_graphs)
        for a, b in zip(kv, r[4]):
            assert torch.equal(a, b), (mode, use_graphs)
    assert len(set(r[2][0, 3:].tolist())) > 3                      # a real decode, not one token repeated

    # ---- (d) TWO, FOUR and EIGHT utterances (round 5: the one-launch step serves groups of up to eight rows -- three more kernels): the
    # launch-per-kernel form and the one-launch step at full depth, eager and replayed, and (two and four rows) each utterance's tokens are the
    # ones it gets alone (rows do not see each other) -------------------------------------------------------------------------------------
    mel8 = torch.cat([mel] + [synthetic_mel(1, 3000, 80, 777 + k).cuda() for k in range(7)])
    xa8 = enc.get_audio_features(mel8)
    for rows in (2, 4, 8):
        xa_r = xa8[:rows].contiguous()
        outs2 = []
        for mode in (0, 2):
            lib.wm_set_decode_chain(mode)
            dec = WhisperDecoding(Path(out), options=DecodingOptions(sample_len=10))
            dec.detect_language(xa_r)
            for use_graphs in (False, True):
                dec.use_graphs = use_graphs
                for st in dec._state.values():
                    st['graphs'].clear()
                before = native.chain_status()["launches"]
                t, lp, _ = dec.main_loop(xa_r, ignore_eot=True)
                assert (native.chain_status()["launches"] > before) == (mode > 0), (rows, mode, use_graphs)
                outs2.append((mode, use_graphs, t.cpu(), lp.cpu(), [c.clone() for c in dec._state[rows]['kv']]))
            del dec
        for mode, use_graphs, t, lp, kv in outs2[1:]:
            assert torch.equal(t, outs2[0][2]) and torch.equal(lp, outs2[0][3]), (rows, mode, use_graphs)
            for a, b in zip(kv, outs2[0][4]):
                assert torch.equal(a, b), (rows, mode, use_graphs)
        if rows <= 4:
            assert torch.equal(outs2[0][2][0], r[2][0])                 # utterance 0 beside others == utterance 0 alone
        assert not torch.equal(outs2[0][3][0], outs2[0][3][1])
        st_ = native.chain_status()
        assert not st_["error_pending"] and not st_["declined"], (rows, st_)


# ------------------------------------------------------------------------------------------ two rows in one launch
@pytest.mark.parametrize("model,weight_only,int8_kv", [("micro-fullvocab", False, False), ("micro-fullvocab", True, True),
                                                        ("tiny", True, False), ("tiny", "int4", True), ("large-v2-6layer", True, True),
                                                        ("large-v2-6layer", False, False)])
def test_two_row_step_equals_the_launch_per_kernel_path(lib, tmp_path_factory, chain_rearmed, model, weight_only, int8_kv):
    """Round 5: groups of TWO rows (two utterances decoded together) take the one-launch forms too -- 2 x (heads + heads x pieces)
    attention workgroups, the Linear stages carry both rows in the MFMA's A operand.  Same arithmetic as a launch per kernel: token
    ids, log-probabilities and the whole KV cache of a batch-2 decode are IDENTICAL in the three forms, eagerly and under graph
    replay, no workgroup gave up a wait, and each utterance's tokens are the ones it gets when it is decoded alone."""
    from test_gpu_model import build_engine
    if model not in synthetic.DIMS:
        pytest.skip(f"no synthetic model {model}")
    tmp = tmp_path_factory.mktemp("chain2")
    dims = Dims(**synthetic.DIMS[model])
    kv_scales = [0.05 + 0.01 * i for i in range(dims.n_text_layer)] if int8_kv else None
    eng = build_engine(tmp, model, 3, weight_only, int8_kv, kv_scales)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(2, 2 * dims.n_audio_ctx, dims.n_mels, 81).cuda())
    outs = []
    for on in (0, 1, 2):
        lib.wm_set_decode_chain(on)
        dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=12))
        dec.detect_language(xa)
        for use_graphs in (False, True):
            dec.use_graphs = use_graphs
            for st in dec._state.values():
                st['graphs'].clear()
            before = native.chain_status()["launches"]
            t, lp, _ = dec.main_loop(xa, ignore_eot=True)
            assert (native.chain_status()["launches"] > before) == (on > 0), (on, use_graphs)
            outs.append((on, use_graphs, t.cpu(), lp.cpu(), [c.clone() for c in dec._state[2]['kv']]))
        del dec
    st_ = native.chain_status()
    assert not st_["error_pending"] and not st_["declined"], st_
    ref = outs[0]
    for on, use_graphs, t, lp, kv in outs[1:]:
        assert torch.equal(t, ref[2]), (on, use_graphs)
        assert torch.equal(lp, ref[3]), (on, use_graphs)
        for a, b in zip(kv, ref[4]):
            assert torch.equal(a, b), (on, use_graphs)
    assert len(set(ref[2][0, 3:].tolist())) > 3 and not torch.equal(ref[2][0], ref[2][1])
    lib.wm_set_decode_chain(2)
    for b in (0, 1):                                                           # alone: the one-row launch
        solo = WhisperDecoding(eng, options=DecodingOptions(sample_len=12))
        xb = xa[b:b + 1].contiguous()
        solo.detect_language(xb)
        t1, lp1, _ = solo.main_loop(xb, ignore_eot=True)
        assert torch.equal(t1[0].cpu(), ref[2][b]) and torch.equal(lp1[0].cpu(), ref[3][b]), b
        del solo


@pytest.mark.parametrize("rows", [3, 4, 5, 7, 8])
@pytest.mark.parametrize("model,weight_only,int8_kv", [("micro-fullvocab", False, False), ("micro-fullvocab", True, True), ("tiny", True, False),
                                                        ("tiny", "int4", True), ("large-v2-6layer", True, True), ("large-v2-6layer", False, False)])
def test_three_to_eight_row_steps_equal_the_launch_per_kernel_path(lib, tmp_path_factory, chain_rearmed, model, weight_only, int8_kv, rows):
    """Round 5: groups of THREE to EIGHT rows take the one-launch forms (gemv_chain_kernel<.., NR = 4 | 8>: the Linear stages carry the
    rows in the MFMA's A operand, the cross-attention's (row, head, piece) items run two per workgroup -- in two rounds at 7 and 8 rows
    of a 20-head model -- with their K / V rows in registers).
    Token ids, log-probabilities and the whole KV cache are IDENTICAL to a launch per kernel in the three forms, eagerly and under
    graph replay; no workgroup gave up a wait; up to four rows each utterance's tokens are the ones it gets alone.  (4-bit weights keep the
    launch-per-kernel path from three rows on: asserted too.)"""
    from test_gpu_model import build_engine
    if model not in synthetic.DIMS:
        pytest.skip(f"no synthetic model {model}")
    tmp = tmp_path_factory.mktemp(f"chain{rows}")
    dims = Dims(**synthetic.DIMS[model])
    kv_scales = [0.05 + 0.01 * i for i in range(dims.n_text_layer)] if int8_kv else None
    eng = build_engine(tmp, model, 3, weight_only, int8_kv, kv_scales)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(rows, 2 * dims.n_audio_ctx, dims.n_mels, 83).cuda())
    takes_chain = weight_only != "int4"
    outs = []
    for on in (0, 1, 2):
        lib.wm_set_decode_chain(on)
        dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=12))
        dec.detect_language(xa)
        for use_graphs in (False, True):
            dec.use_graphs = use_graphs
            for st in dec._state.values():
                st['graphs'].clear()
            before = native.chain_status()["launches"]
            t, lp, _ = dec.main_loop(xa, ignore_eot=True)
            assert (native.chain_status()["launches"] > before) == (on > 0 and takes_chain), (on, use_graphs)
            outs.append((on, use_graphs, t.cpu(), lp.cpu(), [c.clone() for c in dec._state[rows]['kv']]))
        del dec
    st_ = native.chain_status()
    assert not st_["error_pending"] and not st_["declined"], st_
    ref = outs[0]
    for on, use_graphs, t, lp, kv in outs[1:]:
        assert torch.equal(t, ref[2]), (on, use_graphs)
        assert torch.equal(lp, ref[3]), (on, use_graphs)
        for a, b in zip(kv, ref[4]):
            assert torch.equal(a, b), (on, use_graphs)
    assert len(set(ref[2][0, 3:].tolist())) > 3 and not torch.equal(ref[3][0], ref[3][rows - 1])      # (different utterances: different log-probabilities)
    lib.wm_set_decode_chain(2)
    # alone: the one-row launch.  (Up to four rows: from five on the 3-token prefill -- 3 x rows activation rows -- and the vocabulary projection
    # leave the <= 16-row kernels whose per-row arithmetic does not depend on the batch, on EVERY path: a different class)
    for b in ((0, rows - 1) if rows <= 4 else ()):
        solo = WhisperDecoding(eng, options=DecodingOptions(sample_len=12))
        xb = xa[b:b + 1].contiguous()
        solo.detect_language(xb)
        t1, lp1, _ = solo.main_loop(xb, ignore_eot=True)
        assert torch.equal(t1[0].cpu(), ref[2][b]) and torch.equal(lp1[0].cpu(), ref[3][b]), b
        del solo


# ------------------------------------------------------------------------------------------ hardening
def test_one_launch_step_is_declined_on_a_cu_masked_stream(lib, tmpdir_module, chain_rearmed):
    """A stream whose CU mask leaves it fewer CUs than the device has cannot hold the step's workgroups together: the library looks
    at the mask of the stream of the call and takes the launch-per-kernel path there (no give-up, no second of spinning) -- same
    bits as on an ordinary stream."""
    eng, dims = _small_engine(tmpdir_module)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(1, 2 * dims.n_audio_ctx, dims.n_mels, 81).cuda())
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    masked = native.create_masked_stream([i < n_cu - 64 for i in range(n_cu)], 0)
    full = native.create_masked_stream([True] * n_cu, 0)
    res = {}
    for name, strm in (("full", full), ("masked", masked), ("ordinary", torch.cuda.Stream())):
        dec = WhisperDecoding(eng)
        cfg = dec.decoder_config
        cap, V = cfg['num_text_ctx'], cfg['vocab_size']
        st = dec._fast_state(1, xa.device)
        cross = dec._cross_persistent(xa, st)
        tokens = torch.tensor([[50258, 50259, 50359, 1000, 2000]], dtype=torch.int32, device="cuda")
        logits = torch.empty((1, 3, V), dtype=torch.float16, device="cuda")
        torch.cuda.synchronize()
        before = native.chain_status()
        with torch.cuda.stream(strm):
            dec.decoder_session.decoder_step(tokens[:, :3], dec.positional_embedding[0:3], cross, None, cap, st['kv'], cap, logits, 0, strm.cuda_stream)
            l1 = torch.empty((1, 1, V), dtype=torch.float16, device="cuda")
            dec.decoder_session.decoder_step(tokens[:, 3:4], dec.positional_embedding[3:4], cross, st['kv'], cap, st['kv'], cap, l1, 3, strm.cuda_stream)
        torch.cuda.synchronize()
        after = native.chain_status()
        res[name] = (l1.clone(), after["launches"] - before["launches"], after["declined_calls"] - before["declined_calls"])
        del dec, st
    assert res["full"][1] == 1 and res["ordinary"][1] == 1 and res["masked"][1] == 0, {k: v[1:] for k, v in res.items()}
    assert res["masked"][2] == 1
    assert torch.equal(res["full"][0].view(torch.int16), res["masked"][0].view(torch.int16))
    assert torch.equal(res["full"][0].view(torch.int16), res["ordinary"][0].view(torch.int16))
    st_ = native.chain_status()
    assert not st_["error_pending"] and not st_["declined"]


def test_a_step_that_gives_up_is_decoded_again_on_the_launch_per_kernel_path(lib, tmpdir_module, chain_rearmed, caplog):
    """Another tenant holds the LDS of half the CUs while the one-launch step is dispatched (wm_debug_occupy: 128 workgroups of 100 KB
    that sleep for four seconds): the step's workgroups are not resident together, the ones that run give up after their bounded
    waits (about a second) and set the error word.  `main_loop` notices, warns ONCE, and decodes the utterance again in the same
    process on the launch-per-kernel path -- the tokens, log-probabilities and cache of an undisturbed run -- and the device stays
    off the one-launch forms until wm_set_decode_chain re-arms it.  A caller of the C ABI that does not look gets rc != 0 from its
    next wm_decoder_step."""
    eng, dims = _small_engine(tmpdir_module)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(1, 2 * dims.n_audio_ctx, dims.n_mels, 81).cuda())
    dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=10))
    dec.detect_language(xa)
    t_ref, lp_ref, _ = dec.main_loop(xa, ignore_eot=True)
    kv_ref = [c.clone() for c in dec._state[1]['kv']]
    assert native.chain_status()["launches"] > 0 and not native.chain_status()["declined"]
    del dec

    # (1) through the product loop
    WhisperDecoding._chain_warned = False
    dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=10))
    dec.graph_prefill = False           # (round 6: capturing the prefill synchronises the device -- it would wait the occupying workgroups out)
    dec.detect_language(xa)
    torch.cuda.synchronize()
    hog = torch.cuda.Stream()
    native.check(lib.wm_debug_occupy(128, 100 * 1024, 4_000_000, hog.cuda_stream))
    with caplog.at_level(logging.WARNING, logger="whisper_mi355"):
        t, lp, _ = dec.main_loop(xa, ignore_eot=True)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), t_ref.cpu()) and torch.equal(lp.cpu(), lp_ref.cpu())
    for a, b in zip(dec._state[1]['kv'], kv_ref):
        assert torch.equal(a, b)
    st_ = native.chain_status()
    assert st_["declined"] and not st_["error_pending"] and "gave up" in st_["reason"], st_
    assert sum("one-launch decode step gave up" in r.getMessage() for r in caplog.records) == 1
    before = st_["launches"]
    t2, _, _ = dec.main_loop(xa, ignore_eot=True)                   # declined: a launch per kernel, no chain launch any more
    assert torch.equal(t2.cpu(), t_ref.cpu()) and native.chain_status()["launches"] == before
    del dec

    # (2) a C-ABI caller that never asks: the next call fails loudly until the give-up has been acknowledged
    lib.wm_set_decode_chain(-1)                                    # re-armed
    assert not native.chain_status()["declined"]
    dec = WhisperDecoding(eng)
    cfg = dec.decoder_config
    cap, V = cfg['num_text_ctx'], cfg['vocab_size']
    st = dec._fast_state(1, xa.device)
    cross = dec._cross_persistent(xa, st)
    tokens = torch.tensor([[50258, 50259, 50359, 1000, 2000]], dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    l3, l1 = torch.empty((1, 3, V), dtype=torch.float16, device="cuda"), torch.empty((1, 1, V), dtype=torch.float16, device="cuda")
    dec.decoder_session.decoder_step(tokens[:, :3], dec.positional_embedding[0:3], cross, None, cap, st['kv'], cap, l3, 0, s)
    torch.cuda.synchronize()
    native.check(lib.wm_debug_occupy(128, 100 * 1024, 4_000_000, hog.cuda_stream))
    dec.decoder_session.decoder_step(tokens[:, 3:4], dec.positional_embedding[3:4], cross, st['kv'], cap, st['kv'], cap, l1, 3, s)
    torch.cuda.synchronize()
    assert native.chain_status()["error_pending"]
    with pytest.raises(native.WmError, match="gave up"):
        dec.decoder_session.decoder_step(tokens[:, 4:5], dec.positional_embedding[4:5], cross, st['kv'], cap, st['kv'], cap, l1, 4, s)
    err = C.c_int(0)
    native.check(lib.wm_decode_chain_error(C.byref(err)))
    assert err.value != 0
    dec.decoder_session.decoder_step(tokens[:, 3:4], dec.positional_embedding[3:4], cross, st['kv'], cap, st['kv'], cap, l1, 3, s)      # acknowledged: works, a launch per kernel
    torch.cuda.synchronize()
    assert bool(torch.isfinite(l1.float()).all())


@pytest.mark.parametrize("vouched", [True, False])
def test_the_decoder_workspace_needs_no_initialisation(lib, tmpdir_module, chain_rearmed, vouched):
    """The one-launch step keeps a call counter, tagged granules and a pointer table in the caller's workspace.  The caller hands the
    workspace over as the allocator left it (INTEGRATION.md's stub: torch.empty); here it is filled with the worst garbage there is
    -- the bytes a previous life of the SAME state left behind, at a later generation, and 0xFF / 0x00 patterns -- and given a new
    identity (`vouched`) or none (workspace_id 0: the library re-initialises on every call).  Tokens, log-probabilities and cache of
    the first, clean run come out every time, eagerly and replayed."""
    eng, dims = _small_engine(tmpdir_module)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(1, 2 * dims.n_audio_ctx, dims.n_mels, 81).cuda())
    dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=10))
    dec.detect_language(xa)
    t_ref, lp_ref, _ = dec.main_loop(xa, ignore_eot=True)
    kv_ref = [c.clone() for c in dec._state[1]['kv']]
    import session as S
    stale = {k: w.clone() for k, w in dec.decoder_session._workspaces.items()}          # the state as 10 + calls have left it
    for fill in ("stale", 0xFF, 0x00, "random"):
        for k, w in dec.decoder_session._workspaces.items():
            if fill == "stale":
                w.copy_(stale[k])
            elif fill == "random":
                w.copy_(torch.randint(0, 256, (w.numel(),), dtype=torch.uint8, device=w.device))
            else:
                w.fill_(fill)
            w._wm_id = next(S._WORKSPACE_IDS) if vouched else 0
        torch.cuda.synchronize()
        for use_graphs in (False, True):
            dec.use_graphs = use_graphs
            for st in dec._state.values():
                st['graphs'].clear()
            dec.detect_language(xa)
            t, lp, _ = dec.main_loop(xa, ignore_eot=True)
            assert torch.equal(t.cpu(), t_ref.cpu()) and torch.equal(lp.cpu(), lp_ref.cpu()), (fill, use_graphs)
            for a, b in zip(dec._state[1]['kv'], kv_ref):
                assert torch.equal(a, b), (fill, use_graphs)
    st_ = native.chain_status()
    assert st_["launches"] > 0 and not st_["error_pending"] and not st_["declined"], st_


def test_a_first_call_under_stream_capture_carries_its_own_initialisation(lib, tmpdir_module, chain_rearmed):
    """The library remembers which workspaces it has initialised -- but a call that is being CAPTURED enqueues nothing, so it must not
    be remembered: a decoder step captured as the very first call on a workspace, replayed, then followed by an eager call with the
    same pointers and workspace id, reads a table and granules that exist (before: the eager call skipped the table write and the
    launch read null pointers)."""
    eng, dims = _small_engine(tmpdir_module)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(1, 2 * dims.n_audio_ctx, dims.n_mels, 81).cuda())
    outs = []
    for captured_first in (False, True):
        dec = WhisperDecoding(eng)
        cfg = dec.decoder_config
        cap, V = cfg['num_text_ctx'], cfg['vocab_size']
        st = dec._fast_state(1, xa.device)
        cross = dec._cross_persistent(xa, st)
        tokens = torch.zeros((1, cap + 1), dtype=torch.int32, device="cuda")
        tokens[0, :6] = torch.tensor([50258, 50259, 50359, 1000, 2000, 3000], dtype=torch.int32)
        l3, l1 = torch.empty((1, 3, V), dtype=torch.float16, device="cuda"), torch.empty((1, 1, V), dtype=torch.float16, device="cuda")
        side = torch.cuda.Stream()
        sess, pos = dec.decoder_session, dec.positional_embedding
        with torch.cuda.stream(side):
            sess.decoder_step(tokens[:, :3], pos[0:3], cross, None, cap, st['kv'], cap, l3, 0, side.cuda_stream)       # (L = 3: another workspace)
            counter = torch.full((1,), 3, dtype=torch.int32, device="cuda")
            got = []
            if captured_first:
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    sess.decoder_step(tokens, pos, cross, st['kv'], cap, st['kv'], cap, l1, 1, side.cuda_stream, n_past_dev=counter, n_new=1)
                    native.check(lib.wm_step_advance(counter.data_ptr(), side.cuda_stream))
                graph.replay()
                got.append(l1.clone())
                graph.replay()
                got.append(l1.clone())
            else:
                for cur in (4, 5):
                    sess.decoder_step(tokens[:, cur - 1:cur], pos[cur - 1:cur], cross, st['kv'], cap, st['kv'], cap, l1, cur - 1, side.cuda_stream)
                    got.append(l1.clone())
            # ... and an eager call on the SAME workspace (same shapes, same id) behind the replays
            sess.decoder_step(tokens, pos, cross, st['kv'], cap, st['kv'], cap, l1, 1, side.cuda_stream,
                              n_past_dev=torch.full((1,), 5, dtype=torch.int32, device="cuda"), n_new=1)
            got.append(l1.clone())
        torch.cuda.synchronize()
        outs.append(got)
        del dec, st
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    st_ = native.chain_status()
    assert not st_["error_pending"] and not st_["declined"], st_


# ------------------------------------------------------------------------------------------ the encoder beside the loop gives its CUs back
def test_prefetched_encoder_pass_is_bit_identical_whenever_the_budget_is_released(lib, tmpdir_module):
    """WhisperEncoding.prefetch issues the pass layer by layer (wm_encoder_forward_range) on a CU budget and hands the layers still to
    come to the whole chip once `loop_ended()`'s event has passed: whatever the cut -- released before the first layer, somewhere in
    the middle, never -- the audio features are the bits of the plain pass (same tiles, other workgroups), and the pieces of a pass
    issued by hand add up to it."""
    import time
    eng, dims = _small_engine(tmpdir_module, "tiny", True, False)
    enc = WhisperEncoding(eng)
    enc.prefetch_min_batch = 1          # (round 6: batches of up to eight clips are not run beside the loop by default; the helper path is what is under test)
    mel = synthetic_mel(6, 2 * dims.n_audio_ctx, dims.n_mels, 5).cuda()
    ref = enc.get_audio_features(mel).clone()
    released = []
    for when in ("at once", "later", "never"):
        enc.prefetch(mel, 64)
        if when == "at once":
            enc.loop_ended()
        elif when == "later":
            time.sleep(0.02)
            torch.cuda.synchronize()
            enc.loop_ended()
        xa = enc.collect()
        torch.cuda.synchronize()
        released.append(enc.last_release_layer)
        assert torch.equal(xa.view(torch.int16), ref.view(torch.int16)), when
    assert released[0] is not None and released[0] <= 2, released
    # by hand: three ranges with three budgets
    out = torch.empty_like(ref)
    s = torch.cuda.current_stream().cuda_stream
    n = dims.n_audio_layer
    enc.session.encoder_forward_range(mel, out, s, 32, 0, 1)
    enc.session.encoder_forward_range(mel, out, s, 0, 1, n - 1)
    enc.session.encoder_forward_range(mel, out, s, 96, n - 1, n)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    with pytest.raises(native.WmError):
        enc.session.encoder_forward_range(mel, out, s, 0, 2, 1)


@pytest.mark.parametrize("rows,int8_kv", [(1, False), (2, True), (4, False), (8, False), (8, True)])
def test_one_launch_step_over_a_long_decode(lib, tmpdir_module, chain_rearmed, rows, int8_kv):
    """The one-launch step from an empty cache to the end of the decoder's context (n_text_ctx = 448 positions): the self-attention stage
    takes its cached rows from LDS (set out a layer ahead) while they fit the kernel's dynamic LDS -- 96 KB at one and two rows, 64 KB at
    3-4, 24 KB at 5-8 rows: 384 / 255 / 92 positions with an fp16 cache -- and from memory beyond: every kernel crosses its limit here.
    Tokens and log-probabilities of all ~ 440 steps equal the launch-per-kernel path's."""
    eng, dims = _small_engine(tmpdir_module, "tiny", True, int8_kv)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(rows, 2 * dims.n_audio_ctx, dims.n_mels, 85).cuda())
    n = dims.n_text_ctx - 8
    outs = {}
    for mode in (0, 2):
        lib.wm_set_decode_chain(mode)
        dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=n))
        dec.detect_language(xa)
        before = native.chain_status()["launches"]
        t, lp, _ = dec.main_loop(xa, ignore_eot=True)
        assert (native.chain_status()["launches"] > before) == (mode > 0)
        assert t.shape[1] >= n
        outs[mode] = (t.cpu(), lp.cpu(), [c.clone() for c in dec._state[rows]['kv']])
        del dec
    st_ = native.chain_status()
    assert not st_["error_pending"] and not st_["declined"], st_
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    for a, b in zip(outs[0][2], outs[2][2]):
        assert torch.equal(a, b)


def test_graph_capture_survives_an_encoder_pass_on_the_helper_thread(lib, tmpdir_module, chain_rearmed):
    """The pipelined schedule captures the decode step's graphs while the NEXT batch's encoder is being issued layer by layer by
    WhisperEncoding.prefetch's helper thread (event queries, launches, allocations on its own stream).  Under the default "global"
    capture mode a runtime call from that thread inside the capture window invalidated the capture (hipErrorStreamCaptureInvalidated: seen
    once in ~ 25 bench runs).  Forty fresh captures, each beside a pass in flight (a small model: the helper is in the runtime almost all
    the time): every loop returns the tokens of the run without a pass beside it, every pass the features of the plain call."""
    eng, dims = _small_engine(tmpdir_module, "tiny", True, False)
    enc = WhisperEncoding(eng)
    mel = synthetic_mel(12, 2 * dims.n_audio_ctx, dims.n_mels, 5).cuda()
    xa = enc.get_audio_features(mel).clone()
    lib.wm_set_decode_chain(0)                                     # (12 rows: a launch per kernel anyway; the graphs are what is under test)
    dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=6))
    dec.detect_language(xa)
    t_ref, lp_ref, _ = dec.main_loop(xa, ignore_eot=True)
    t_ref, lp_ref = t_ref.cpu(), lp_ref.cpu()
    for it in range(40):
        for st in dec._state.values():
            st['graphs'].clear()                                   # capture again
        enc.prefetch(mel, 64)
        t, lp, _ = dec.main_loop(xa, ignore_eot=True)
        enc.loop_ended()
        xa2 = enc.collect()
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), t_ref) and torch.equal(lp.cpu(), lp_ref), it
        assert torch.equal(xa2.view(torch.int16), xa.view(torch.int16)), it
    # ... and beside a thread that does NOT take the library's lock and never leaves the runtime (event record / synchronize / query, a
    # small allocation and a kernel on its own stream): what protects the capture here is its thread-local error mode alone
    import threading
    stop, errors = threading.Event(), []

    def hammer():
        try:
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                while not stop.is_set():
                    x = torch.empty(257, device="cuda").fill_(1.0)
                    ev = torch.cuda.Event(); ev.record()
                    ev.synchronize()
                    assert ev.query() and float(x[0]) == 1.0
        except BaseException as exc:                                   # noqa: BLE001
            errors.append(exc)
    th = threading.Thread(target=hammer, daemon=True)
    th.start()
    try:
        for it in range(30):
            for st in dec._state.values():
                st['graphs'].clear()
            t, lp, _ = dec.main_loop(xa, ignore_eot=True)
            assert torch.equal(t.cpu(), t_ref) and torch.equal(lp.cpu(), lp_ref), it
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    del dec


@pytest.mark.parametrize("rows", [2, 4, 7])
def test_multi_row_step_with_rows_that_finish_at_different_times(lib, tmpdir_module, chain_rearmed, rows):
    """Per-row completion in a group of two / four / seven rows (the three multi-row kernels): the rows end (their `row_limit`) at different
    steps.  Both forms drop a finished row from the attention work (the one-launch step reads the list of live rows at the head of every
    launch, under graph replay too) -- tokens and log-probabilities of every row are the same in both forms and, up to four rows, equal to
    what each utterance gets alone with its own limit."""
    eng, dims = _small_engine(tmpdir_module)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(rows, 2 * dims.n_audio_ctx, dims.n_mels, 81).cuda())
    limits = torch.tensor([3, 14, 7, 1, 12, 5, 9][:rows], dtype=torch.int32)
    outs = {}
    for mode in (0, 2):
        lib.wm_set_decode_chain(mode)
        dec = WhisperDecoding(eng, options=DecodingOptions(sample_len=16))
        dec.detect_language(xa)
        before = native.chain_status()["launches"]
        t, lp, _ = dec.main_loop(xa, row_limit=limits)
        assert (native.chain_status()["launches"] > before) == (mode > 0)
        outs[mode] = (t.cpu(), lp.cpu())
        del dec
    st_ = native.chain_status()
    assert not st_["error_pending"] and not st_["declined"], st_
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    eot = 50257
    t = outs[2][0]
    for b in range(rows):
        assert int((t[b, 3:] != eot).sum()) == int(limits[b]), b
    for b in (range(rows) if rows <= 4 else ()):
        solo = WhisperDecoding(eng, options=DecodingOptions(sample_len=16))
        xb = xa[b:b + 1].contiguous()
        solo.detect_language(xb)
        t1, lp1, _ = solo.main_loop(xb, row_limit=limits[b:b + 1])
        n = t1.shape[1]
        assert torch.equal(t1[0].cpu(), t[b, :n]) and bool((t[b, n:] == eot).all())
        assert torch.equal(lp1[0].cpu(), outs[2][1][b])
        del solo


@pytest.mark.parametrize("best_of", [2, 5])
def test_best_of_candidates_of_one_utterance_take_the_multi_row_step(lib, tmpdir_module, chain_rearmed, best_of):
    """`best_of = 2 | 5` (the reference's default when sampling: W/decoding.py:113) on ONE utterance: candidate rows that share the clip's audio (their own copies of its
    cross K/V) -- a group of two / five rows, i.e. the two-row / 5-8-row one-launch step with the draw inside the greedy kernel.  The run is repeatable under torch.manual_seed,
    identical to the launch-per-kernel path (the draws are keyed on seed, global row, position and token: not on the decode form), the
    candidates differ from each other, and the language pass and the loop share ONE buffer set (no re-allocation between them)."""
    eng, dims = _small_engine(tmpdir_module, "micro-fullvocab", True, True)
    enc = WhisperEncoding(eng)
    xa = enc.get_audio_features(synthetic_mel(1, 2 * dims.n_audio_ctx, dims.n_mels, 79).cuda())
    outs = {}
    for mode in (0, 2):
        lib.wm_set_decode_chain(mode)
        dec = WhisperDecoding(eng, options=DecodingOptions(temperature=0.8, best_of=best_of, sample_len=10))
        languages, _ = dec.detect_language(xa)
        assert len(languages) == 1 and list(dec._state.keys()) == [best_of]            # the pass ran over the candidates' rows: the loop's buffer set
        state = dec._state[best_of]
        before = native.chain_status()["launches"]
        torch.manual_seed(11)
        t, lp, nsp = dec.main_loop(xa)
        assert dec._state[best_of] is state and t.shape[0] == best_of
        assert (native.chain_status()["launches"] > before) == (mode > 0)
        torch.manual_seed(11)
        t_again, lp_again, _ = dec.main_loop(xa)
        assert torch.equal(t.cpu(), t_again.cpu()) and torch.equal(lp.cpu(), lp_again.cpu())
        res = dec.post_process(t, lp, nsp, xa, languages)
        assert len(res) == 1 and np.isfinite(res[0].avg_logprob)
        outs[mode] = (t.cpu(), lp.cpu())
        del dec
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    assert not torch.equal(outs[2][0][0], outs[2][0][1])
