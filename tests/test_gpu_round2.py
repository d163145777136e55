"""Parity rows closed in round 2 (all through the C ABI, on a real MI355X):

  * f3 -- the calibration tool (`torch_whisper_convert.capture_kv_activation_range` = the amax hook of the decode
    self-attention kernel) against the oracle's restatement of W/smoothquant.py:117-175 on the SAME token path;
  * a fixture the reference itself produced at a real Whisper shape (tests/golden/model_tiny_en_shape.npz:
    384 wide, 6 heads, 1500 audio positions, 51 864 tokens) held against the engine;
  * BASELINE.json configs[1] and [2] (fp16, weight-only int8) at large-v2 WIDTH against the oracle;
  * two batches encoded into one output buffer do not share cross K/V (the cache is keyed on encoder runs);
  * `bench.py --gpus 1 --force-dist`: RCCL initialisation + scatter / gather on one rank.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import synthetic  # noqa: E402
import torch_whisper_convert as TWC  # noqa: E402
from decoding import WhisperDecoding  # noqa: E402
from encoding import WhisperEncoding  # noqa: E402
from oracle.whisper_oracle import (Dims, OracleConfig, OracleModel, greedy_reference_run, kv_amax_on_token_path,
                                   synthetic_mel, synthetic_state_dict)  # noqa: E402
from test_gpu_model import LOGIT_TOL, LOGIT_TOL_INT8_KV, _engine_vs_oracle, _flat, build_engine  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tmpdir_module(tmp_path_factory):
    return str(tmp_path_factory.mktemp("engines2"))


# ---------------------------------------------------------------------------------------------- f3: calibration tool
@pytest.mark.parametrize("name,dims_d,seed", [
    ("micro-fullvocab", None, 3),
    ("large-v2-2layer-cal", dict(synthetic.DIMS["large-v2"], n_audio_layer=2, n_text_layer=2), 12),
])
def test_calibration_tool_matches_oracle(tmpdir_module, name, dims_d, seed):
    """`torch_whisper_convert.py -kv`: the per-layer statistic max(|q|,|k|,|v|) the decode self-attention kernel keeps
    (atomicMax hook, csrc/attn_decode.hip) while the fp16 engine runs the language-ID pass and the greedy loop, against
    the oracle's hooks (W/smoothquant.py:117-175; scale = amax / 127, W/utils/convert.py:76-78) teacher-forced with the
    tokens the engine decoded.  The values are maxima of fp16 numbers, so they agree to fp16 rounding of sums taken in
    different orders: 2 fp16 ulps (2^-9 relative), and the .bin files hold exactly amax / 127 in fp32."""
    if dims_d is not None:
        synthetic.DIMS[name] = dims_d
    dims = Dims(**synthetic.DIMS[name])
    eng = build_engine(tmpdir_module, name, seed)
    n_clips, n_tok = 4, 6
    mels = synthetic_mel(n_clips, 2 * dims.n_audio_ctx, dims.n_mels, 4321)
    log = []
    amax = TWC.capture_kv_activation_range(eng, mels, batch=n_clips, sample_len=n_tok, ignore_eot=True, token_log=log)
    assert len(amax) == dims.n_text_layer and all(a > 0 for a in amax)
    (tokens, L0, sot), = log
    assert tokens.shape == (n_clips, L0 + n_tok)
    passes = [[torch.full((n_clips, 1), sot, dtype=torch.long)]]                  # language-ID pass: <|sot|> alone, empty cache
    passes.append([tokens[:, :L0]] + [tokens[:, L0 + j:L0 + j + 1] for j in range(n_tok - 1)])   # the last sampled token is never fed
    oracle = OracleModel(dims, synthetic_state_dict(dims, seed), OracleConfig(act="float16"))
    want = kv_amax_on_token_path(oracle, mels, passes)
    rel = [abs(a - w) / w for a, w in zip(amax, want)]
    print(f"{name}: engine amax {np.round(amax, 4).tolist()} oracle {np.round(want, 4).tolist()} rel {np.round(rel, 5).tolist()}")
    assert max(rel) < 2 ** -9, rel
    # the files build.py --int8_kv_cache reads hold amax / 127 (fp32), one per layer, under the reference's names
    qdir = TWC.write_kv_scales(os.path.join(tmpdir_module, f"quantize_{name}"), amax, {"source": "test"})
    for i, a in enumerate(amax):
        t = np.fromfile(qdir / f"model.decoder.blocks.{i}.attn.query_key_value.scale_y_quant_orig.bin", dtype=np.float32)
        assert t.shape == (1,) and t[0] == np.float32(a) / np.float32(127.0)
    # and an int8-KV engine built from those files runs with exactly those scales
    eng8 = build_engine(tmpdir_module, name, seed, int8_kv=True, kv_scales=[float(np.float32(a) / np.float32(127.0)) for a in amax])
    dec8 = WhisperDecoding(eng8)
    assert dec8.use_int8_kv_cache


def test_calibration_cli_reads_flac(tmpdir_module, golden_dir, tmp_path):
    """The CLI on the reference's kind of input (a LibriSpeech directory of .flac files): wm_flac_decode + wm_log_mel
    feed the calibration, the reference-named files come out."""
    import shutil
    ds = tmp_path / "LibriSpeech" / "valid-clean" / "1089" / "134691"
    ds.mkdir(parents=True)
    for i in range(2):
        shutil.copy(os.path.join(golden_dir, "librispeech_1089-134691-0000.flac"), ds / f"1089-134691-000{i}.flac")
    eng = build_engine(tmpdir_module, "tiny.en", 21)
    out = tmp_path / "quantize"
    args = TWC.parse_arguments(["-o", str(out), "-kv", "--synthetic", "tiny.en", "--seed", "21", "--engine_dir", str(eng),
                                "--dataset_dir", str(tmp_path / "LibriSpeech" / "valid-clean")])
    mels = TWC.load_calibration_mels(args, synthetic.DIMS["tiny.en"])
    assert tuple(mels.shape) == (2, 80, 3000) and mels.dtype == torch.float16 and torch.equal(mels[0], mels[1])
    TWC.run_conversion(args)
    files = sorted(p.name for p in (out / "1-gpu").iterdir())
    assert "config.ini" in files and sum(f.endswith("attn.query_key_value.scale_y_quant_orig.bin") for f in files) == 4


# ---------------------------------------------------------------------------------------------- reference fixture, real shape
def test_tiny_en_shape_engine_matches_reference_golden(tmpdir_module, golden_dir):
    """Engine against outputs of the reference's own PyTorch model (fp16-input mode) at tiny.en width / heads / 1500
    audio positions / gpt2 vocabulary: encoder rows, layer-0 and last-layer cross K/V rows, the reference's top-64
    logits per step (teacher-forced with the reference's ids), greedy ids."""
    fx = np.load(os.path.join(golden_dir, "model_tiny_en_shape.npz"))
    dims_d = {k: int(v) for k, v in zip(fx["dims_keys"], fx["dims"])}
    synthetic.DIMS["tiny.en-2layer"] = dims_d
    dims = Dims(**dims_d)
    eng = build_engine(tmpdir_module, "tiny.en-2layer", int(fx["seed"]))
    enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
    B = int(fx["batch"])
    mel = synthetic_mel(B, 2 * dims.n_audio_ctx, dims.n_mels, int(fx["mel_seed"])).cuda()
    rows = torch.from_numpy(fx["rows"]).cuda()
    xa = enc.get_audio_features(mel)
    d_xa = float((xa[:, rows].float().cpu() - torch.from_numpy(fx["f16_xa"].astype(np.float32))).abs().max())
    cross = dec.xa2cross_key_value(xa)
    d_ckv = 0.0
    for key, layer, kv in (("cross_k0", 0, 0), ("cross_v0", 0, 1), ("cross_vL", -1, 1)):
        got = _flat(cross[layer][:, kv])[:, rows].float().cpu()
        d_ckv = max(d_ckv, float((got - torch.from_numpy(fx[f"f16_{key}"].astype(np.float32))).abs().max()))
    ids = torch.from_numpy(fx["f16_ids"]).cuda()
    top = torch.from_numpy(fx["f16_top_ids"].astype(np.int64)).cuda()
    want_top = torch.from_numpy(fx["f16_top_logits"]).cuda()
    logits, kv = dec.decode(torch.tensor([fx["prompt"].tolist()] * B).cuda(), cross)
    got_ids, worst = [], 0.0
    for s in range(int(fx["n_steps"])):
        last = logits[:, -1].float()
        worst = max(worst, float((last.gather(1, top[:, s]) - want_top[:, s]).abs().max()))
        got_ids.append(last.argmax(-1))
        if s + 1 < int(fx["n_steps"]):
            logits, kv = dec.decode(ids[:, s:s + 1], cross, kv)
    got_ids = torch.stack(got_ids, 1).cpu().numpy()
    print(f"tiny.en shape vs reference: max|xa| {d_xa:.4f}, max|cross K/V| {d_ckv:.4f}, max|top-64 logits| {worst:.4f}")
    assert d_xa < 2e-2 and d_ckv < 2e-2 and worst < LOGIT_TOL, (d_xa, d_ckv, worst)
    safe = fx["f16_margins"] > 2 * LOGIT_TOL
    assert safe.all() and (got_ids == fx["f16_ids"]).all()


# ---------------------------------------------------------------------------------------------- configs[1], [2] at full width
@pytest.mark.parametrize("weight_only,int8_kv,tol", [(False, False, LOGIT_TOL), (True, False, LOGIT_TOL), (False, True, LOGIT_TOL_INT8_KV)])
def test_large_v2_width_all_precisions_match_oracle(tmpdir_module, weight_only, int8_kv, tol):
    """BASELINE.json configs[1] (fp16) and configs[2] (weight-only int8) -- and int8 KV alone -- at large-v2 width
    (1280 wide, 20 heads, 1500 audio positions, 51 865 tokens; 2 + 2 layers so that the CPU oracle finishes in
    seconds).  configs[3] (both) at this width and at full depth: tests/test_gpu_model.py."""
    dims = dict(synthetic.DIMS["large-v2"], n_audio_layer=2, n_text_layer=2)
    d_xa, d_ckv, worst, n_ok, n_safe = _engine_vs_oracle(
        tmpdir_module, dims, "large-v2-2layer", 12, weight_only=weight_only, int8_kv=int8_kv, batch=3, n_steps=4, tol=tol)
    print(f"large-v2 width, weight_only={weight_only} int8_kv={int8_kv}: xa {d_xa:.4f} cross {d_ckv:.4f} logits {worst:.4f} ids {n_ok}/{n_safe}")
    assert d_xa < 3e-2 and d_ckv < 3e-2 and worst < tol, (d_xa, d_ckv, worst)
    assert n_ok == n_safe and n_safe > 0


# ---------------------------------------------------------------------------------------------- cross K/V cache identity
def test_reused_encoder_output_buffer_gets_fresh_cross_kv(tmpdir_module):
    """`get_audio_features_async(mel, out=buf)` invites re-using one output buffer across batches.  The engine writes
    through a raw pointer (torch's version counter does not move), so the cross-K/V cache must key on the encoder RUN:
    a second batch in the same buffer gets its own cross K/V, language pass and tokens."""
    dims = Dims(**synthetic.DIMS["micro-fullvocab"])
    eng = build_engine(tmpdir_module, "micro-fullvocab", 3)
    enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
    dec.sample_len = 6
    mel_a = synthetic_mel(4, 2 * dims.n_audio_ctx, dims.n_mels, 101).cuda()
    mel_b = synthetic_mel(4, 2 * dims.n_audio_ctx, dims.n_mels, 202).cuda()
    buf = torch.empty((4, dims.n_audio_ctx, dims.n_audio_state), dtype=torch.float16, device="cuda")

    def run(mel):
        xa = enc.get_audio_features_async(mel, out=buf)
        assert xa is buf
        cross0 = dec.xa2cross_key_value(xa)[0].clone()
        dec.detect_language(xa)
        fast_cross0 = dec._state[4]['cross'][0].clone()
        tokens, _, _ = dec.main_loop(xa)
        return cross0, fast_cross0, tokens.clone()

    ca, fa, ta = run(mel_a)
    ver = buf._version
    cb, fb, tb = run(mel_b)
    assert buf._version == ver                                    # torch saw no write: the old key would have matched
    assert not torch.equal(ca, cb) and not torch.equal(fa, fb)
    assert torch.equal(ca, fa) and torch.equal(cb, fb)            # by-name engine call == persistent fast-path buffers
    fresh = WhisperDecoding(eng)
    fresh.sample_len = 6
    xb = enc.get_audio_features(mel_b)
    fresh.detect_language(xb)
    t_fresh, _, _ = fresh.main_loop(xb)
    assert torch.equal(tb.cpu(), t_fresh.cpu()) and not torch.equal(ta.cpu(), tb.cpu())
    # within one encoder run the K/V are shared (the reference computes them twice)
    xa = enc.get_audio_features_async(mel_a, out=buf)
    assert dec.xa2cross_key_value(xa)[0].data_ptr() == dec.xa2cross_key_value(xa)[0].data_ptr()
    # an unstamped view of the same memory is never assumed to be known content
    assert dec._features_key(xa[:2]) is None


def test_decoder_state_is_one_buffer_set(tmpdir_module):
    """A second batch size must not keep the first one's KV / cross-K/V / graphs alive (141 GB at B = 576)."""
    dims = Dims(**synthetic.DIMS["micro-fullvocab"])
    eng = build_engine(tmpdir_module, "micro-fullvocab", 3)
    enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
    dec.sample_len = 4
    for n in (6, 5, 6):
        xa = enc.get_audio_features(synthetic_mel(n, 2 * dims.n_audio_ctx, dims.n_mels, n).cuda())
        dec.detect_language(xa)
        t, _, _ = dec.main_loop(xa)
        assert t.shape[0] == n and list(dec._state) == [n]


# ---------------------------------------------------------------------------------------------- bench.py through RCCL, one rank
def test_bench_force_dist_single_rank(tmp_path):
    """`bench.py --gpus 1 --force-dist`: torch.distributed over RCCL initialised, mel scatter and token gather run
    as collectives on one rank, the printed line carries n_gpus == --gpus.  (N > 1 is the driver's to launch: `bench.py
    --gpus N` starts its N ranks itself, tests/test_dp_gloo.py covers the launcher and the N = 2 data path on gloo.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--model", "tiny", "--config", "int8",
           "--batch", "16", "--decode-steps", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
           "--engine-cache", str(tmp_path / "engines")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["batch_per_gpu"] == 16
    assert line["roofline"] is not None and line["roofline"]["bound"] == "hbm"
    out = Path(ROOT) / "gpurun_out"
    if out.is_dir():
        (out / "bench_force_dist.json").write_text(json.dumps(line) + "\n")


# ---------------------------------------------------------------------------------------------- encoder beside a decode loop
def test_encoder_on_a_cu_budget_is_bit_identical(tmpdir_module):
    """wm_encoder_forward_shared: with a budget of CUs the persistent GEMM launches fewer workgroups over the SAME tiles,
    so the audio features are bit-identical to the whole-chip pass -- at large-v2 width (1280: the persistent kernel's
    shapes), for a budget of one workgroup per XCD, a ragged one (rounded down to a multiple of 8) and one above the CU count."""
    synthetic.DIMS["large-v2-2layer"] = dict(synthetic.DIMS["large-v2"], n_audio_layer=2, n_text_layer=2)
    eng = build_engine(tmpdir_module, "large-v2-2layer", 12)
    enc = WhisperEncoding(eng)
    mel = synthetic_mel(3, 3000, 80, 77).cuda()
    whole = enc.get_audio_features(mel)
    assert bool(torch.isfinite(whole.float()).all()) and float(whole.float().abs().max()) > 0
    for budget in (8, 43, 96, 4096):
        part = enc.get_audio_features_async(mel, cu_budget=budget)
        torch.cuda.synchronize()
        assert torch.equal(part, whole), budget
    with pytest.raises(Exception):
        enc.get_audio_features_async(mel, cu_budget=-1)


def test_prefetched_encoder_and_pipelined_evaluation(tmpdir_module):
    """WhisperEncoding.prefetch / collect (the encoder of the next batch enqueued from a helper thread on a stream of its
    own, beside the decode loop) and summarize.eval_engines_stream over three batches: same features, same tokens and
    texts as batch-by-batch evaluation."""
    import summarize as S
    dims = Dims(**synthetic.DIMS["micro-fullvocab"])
    eng = build_engine(tmpdir_module, "micro-fullvocab", 3)
    enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
    enc.prefetch_min_batch = 1          # (round 6: batches of up to eight clips are not run beside the loop by default; the helper path is what is under test)
    dec.sample_len = 6
    mels = [synthetic_mel(4, 2 * dims.n_audio_ctx, dims.n_mels, 300 + i).cuda() for i in range(3)]
    plain = [S.eval_engines(enc, dec, m) for m in mels]
    want_xa = enc.get_audio_features(mels[1])
    enc.prefetch(mels[1], 16)
    with pytest.raises(AssertionError):
        enc.prefetch(mels[2], 16)                                 # one at a time
    got_xa = enc.collect()
    torch.cuda.synchronize()
    assert torch.equal(got_xa, want_xa) and got_xa.wm_generation != want_xa.wm_generation
    for budget in (16, 0):
        piped = list(S.eval_engines_stream(enc, dec, iter(mels), cu_budget=budget))
        assert len(piped) == 3
        for a, b in zip(plain, piped):
            assert [r.tokens for r in a] == [r.tokens for r in b]
            assert [r.text for r in a] == [r.text for r in b]
    assert enc._prefetch is None
    # an error in the helper thread surfaces in collect(), and the encoder is usable afterwards
    enc.prefetch(mels[0], -1)
    with pytest.raises(Exception):
        enc.collect()
    assert enc._prefetch is None
    enc.prefetch(mels[1], 16)
    assert torch.equal(enc.collect(), want_xa)
