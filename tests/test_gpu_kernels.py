"""Kernel-level parity on a real MI355X: every HIP kernel, called through the C ABI, against a
plain fp32 restatement of the same op (and the reference's own tolerances where it has a test).

Tolerances (stated per test): outputs are fp16, so the floor is half an fp16 ulp of the output
magnitude; the reference accepts atol 2e-3 for attention layers (R/tests/test_layer.py:764-768),
5e-3 for the attention plugin incl. int8 KV (R/tests/attention/test_gpt_attention.py:684-693),
2e-2 for fp16 LayerNorm (test_layer_norm.py:74-78), 1.5*colmax/128 per column for weight-only
matmul (R/tests/quantization/_utils.py:66-88) and exact equality for int8 quantise
(test_functional.py:46-50).  We hold the kernels to tighter bounds than those.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import native  # noqa: E402
import weight as W  # noqa: E402
from oracle import decoding_rules as DR  # noqa: E402
from oracle.whisper_oracle import (OracleConfig, OracleModel, Dims, symmetric_quantize_int8, symmetric_quantize_int4, kv_quantize,
                                   kv_dequantize, woq_reference_matmul, woq_colwise_atol)  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return native.load_library()


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def stream():
    return torch.cuda.current_stream().cuda_stream


def rng(seed):
    return np.random.Generator(np.random.Philox(seed))


# ------------------------------------------------------------------------------------------ big GEMM
@pytest.mark.parametrize("M,N,K,w8,act,resid", [
    (200, 256, 192, 0, 0, False), (128, 128, 64, 0, 1, True), (300, 384, 256, 1, 0, False),
    (77, 128, 128, 1, 1, True), (1500, 1280, 1280, 1, 0, True), (1500, 3840, 1280, 0, 0, False),
])
def test_gemm_big(lib, M, N, K, w8, act, resid):
    r = rng(M + N + K)
    A = (r.standard_normal((M, K)) * 0.5).astype(np.float16)
    Wf = (r.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    bias = (r.standard_normal(N) * 0.1).astype(np.float16)
    res = (r.standard_normal((M, N)) * 0.5).astype(np.float16) if resid else None
    if w8:
        q, s = symmetric_quantize_int8(Wf)
        w_dev, s_dev = dev(q), dev(s)
        ref = (A.astype(np.float32) @ q.astype(np.float32).T) * s.astype(np.float32)[None, :]
    else:
        w_dev, s_dev = dev(Wf), None
        ref = A.astype(np.float32) @ Wf.astype(np.float32).T
    ref = (ref + bias.astype(np.float32)).astype(np.float16).astype(np.float32)
    if act == 1:
        ref = torch.nn.functional.gelu(torch.from_numpy(ref)).half().float().numpy()
    if resid:
        ref = (ref + res.astype(np.float32)).astype(np.float16).astype(np.float32)
    a_dev, b_dev = dev(A), dev(bias)
    r_dev = dev(res) if resid else None
    out = torch.zeros((M, N), dtype=torch.float16, device="cuda")
    # w8: the engines' own path for a weight-only matrix of an M >> 16 stage (expansion to fp16(fp16(q) * scale), fp16 MFMA GEMM)
    ws = torch.empty(N * K * 2 if w8 else 256, dtype=torch.uint8, device="cuda")
    native.check(lib.wm_gemm(a_dev.data_ptr(), K, M, K, w_dev.data_ptr(), N, w8,
                             s_dev.data_ptr() if s_dev is not None else None, b_dev.data_ptr(),
                             r_dev.data_ptr() if resid else None, N, act, out.data_ptr(), N, ws.data_ptr(), ws.numel(), stream()))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    # one fp16 ulp of the output magnitude (fp32 accumulation order is the only other difference)
    tol = 2.0 ** -10 * max(1.0, np.abs(ref).max())
    if w8:
        # the product path multiplies fp16(fp16(q) * scale) like the reference kernels (weightOnlyMatrixVectorMultiplication.cu:
        # 44-53); against the reference TEST's ground truth (x @ q) * scale that is the tolerance the reference accepts
        # (R/tests/quantization/_utils.py:66-88), and against the same arithmetic restated it is one ulp
        deq = (q.astype(np.float16) * s[:, None]).astype(np.float16).astype(np.float32)
        ref2 = (A.astype(np.float32) @ deq.T + bias.astype(np.float32)).astype(np.float16).astype(np.float32)
        if act == 1:
            ref2 = torch.nn.functional.gelu(torch.from_numpy(ref2)).half().float().numpy()
        if resid:
            ref2 = (ref2 + res.astype(np.float32)).astype(np.float16).astype(np.float32)
        assert np.abs(got - ref2).max() <= tol, (np.abs(got - ref2).max(), tol)
        tol = max(tol, 1.5 * np.abs(ref).max() / 128.0)
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)


def test_conv1d_gelu_matches_reference_golden(lib, golden_dir):
    """wm_conv1d_gelu against the reference's own Conv1d + GELU outputs (tests/golden/ops.npz: conv1 k3 s1 p1, conv2 k3 s2 p1,
    produced by W/torch_model.py's Conv1d in fp32): the encoder's strided-view GEMM, alone."""
    import os
    ops = np.load(os.path.join(golden_dir, "ops.npz"))
    x = ops["conv_x"]                                           # [2, 8, 20]
    B, Cin, T = x.shape

    def run(x_bct, w, b, stride):
        Bn, Ci, Tn = x_bct.shape
        pad = np.zeros((Bn * (Tn + 2) * Ci + 512,), dtype=np.float16)
        pad[:Bn * (Tn + 2) * Ci].reshape(Bn, Tn + 2, Ci)[:, 1:Tn + 1] = x_bct.transpose(0, 2, 1)
        Cout = w.shape[0]
        wg = np.zeros((128, W.conv_weight_as_gemm(w).shape[1]), dtype=np.float16)      # C_out padded to the GEMM's 128-column tile
        wg[:Cout] = W.conv_weight_as_gemm(w)
        bg = np.zeros(128, dtype=np.float16)
        bg[:Cout] = b
        To = Tn // stride
        out = torch.zeros((Bn, To, 128), dtype=torch.float16, device="cuda")
        xd, wd, bd = dev(pad), dev(wg), dev(bg)
        native.check(lib.wm_conv1d_gelu(xd.data_ptr(), Bn, Tn, Ci, wd.data_ptr(), wg.shape[1], bd.data_ptr(), 128, stride, 1,
                                        out.data_ptr(), stream()))
        torch.cuda.synchronize()
        assert float(out[:, :, Cout:].abs().max()) == 0.0       # GELU(0 . x + 0) = 0 in the padding channels
        return out[:, :, :Cout].float().cpu().numpy().transpose(0, 2, 1)          # back to [B, C, T]

    y1 = run(x.astype(np.float16), ops["conv1_w"], ops["conv1_b"], 1)
    assert np.abs(y1 - ops["conv1_out"]).max() < 4e-3           # fp16 inputs / weights / output against the fp32 reference
    y2 = run(ops["conv1_out"].astype(np.float16), ops["conv2_w"], ops["conv2_b"], 2)
    assert y2.shape == ops["conv2_out"].shape and np.abs(y2 - ops["conv2_out"]).max() < 4e-3


def test_argmax_first_index_wins(lib):
    r = rng(3)
    V = 51865
    lg = (r.standard_normal((5, V)) * 2).astype(np.float16)
    lg[1, 777] = lg[1, 40000] = np.float16(30.0)                # a tie: the first index wins (torch.argmax)
    lg[2, V - 1] = np.float16(31.0)
    lg[3, :] = np.float16(-np.inf); lg[3, 9] = np.float16(-5.0)
    d = dev(lg)
    ids = torch.full((5,), -1, dtype=torch.int32, device="cuda")
    native.check(lib.wm_argmax(d.data_ptr(), V, 5, V, ids.data_ptr(), stream()))
    torch.cuda.synchronize()
    want = torch.from_numpy(lg.astype(np.float32)).argmax(-1).tolist()
    assert ids.tolist() == want and want[1] == 777 and want[2] == V - 1 and want[3] == 9


# --------------------------------------------------------------------------------------- skinny GEMM
def _run_skinny(lib, A, Wmat, w8, ksplit):
    M, K = A.shape
    N = Wmat.shape[0]
    if w8 == 4:
        q, s = symmetric_quantize_int4(Wmat)
        tiles = W.tile_linear_int4(q)
        npad = tiles.shape[0] * 16
        s_dev = dev(np.concatenate([s, np.zeros(npad - N, dtype=np.float16)]))
        ref = (A.astype(np.float32) @ q.astype(np.float32).T) * s.astype(np.float32)[None, :]
    elif w8:
        q, s = symmetric_quantize_int8(Wmat)
        tiles = W.tile_linear(q)
        npad = tiles.shape[0] * 16
        s_pad = np.concatenate([s, np.zeros(npad - N, dtype=np.float16)])
        s_dev = dev(s_pad)
        ref = (A.astype(np.float32) @ q.astype(np.float32).T) * s.astype(np.float32)[None, :]
    else:
        tiles = W.tile_linear(Wmat)
        npad = tiles.shape[0] * 16
        s_dev = None
        ref = A.astype(np.float32) @ Wmat.astype(np.float32).T
    t_dev = dev(tiles.view(np.uint8) if not w8 else tiles.view(np.uint8))
    a_dev = dev(A)
    part = torch.full((ksplit, M, npad), float("nan"), dtype=torch.float32, device="cuda")
    native.check(lib.wm_gemm_skinny(a_dev.data_ptr(), K, M, K, t_dev.data_ptr(), npad // 16, w8,
                                    s_dev.data_ptr() if s_dev is not None else None, ksplit, part.data_ptr(), stream()))
    torch.cuda.synchronize()
    got = part.sum(dim=0).cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got[:, N:]).max(initial=0.0) == 0.0           # padded channels stay zero
    return got[:, :N], ref


@pytest.mark.parametrize("M,N,K,w8,ksplit", [
    (1, 272, 320, 1, 1), (3, 272, 320, 1, 3), (17, 128, 1280, 1, 5), (64, 1280, 1280, 1, 4),
    (1, 272, 320, 0, 1), (5, 200, 256, 0, 2), (33, 128, 1280, 0, 7), (64, 5120, 1280, 0, 3),
    (48, 1280, 5120, 1, 10), (96, 1280, 1280, 1, 4), (128, 3840, 1280, 1, 8), (128, 1280, 5120, 1, 8), (100, 272, 256, 0, 2),
    # packed int4 tiles (K a multiple of 128): M = 1 GEMV, ragged N, every MT variant, split-K
    (1, 272, 384, 4, 1), (3, 272, 384, 4, 3), (17, 128, 1280, 4, 5), (40, 1280, 1280, 4, 2), (64, 3840, 1280, 4, 10),
    (90, 1280, 5120, 4, 8), (128, 5120, 1280, 4, 6),
    # 9 .. 16 row tiles per launch (decode groups of up to 256 utterances)
    (130, 1280, 1280, 1, 8), (192, 3840, 1280, 1, 8), (256, 1280, 5120, 1, 8), (200, 272, 256, 0, 2), (256, 1280, 1280, 0, 4),
    (256, 5120, 1280, 4, 6), (177, 1280, 1280, 4, 5),
])
def test_gemm_skinny(lib, M, N, K, w8, ksplit):
    r = rng(M * 7 + N + K + w8)
    A = (r.standard_normal((M, K)) * 0.5).astype(np.float16)
    Wf = (r.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    got, ref = _run_skinny(lib, A, Wf, w8, ksplit)
    # fp32 accumulation on both sides: only the summation order differs
    assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("w8", [0, 1, 4])
def test_gemm_skinny_rows_do_not_depend_on_the_launch(lib, w8):
    """A row's slabs are the same bits whether it is computed alone (1 MFMA row tile, 4 waves), in a group of 40 or of 130
    (6 / 12 tiles) or of 256 (16 tiles, 70 KB of LDS): the variants differ in schedule, never in arithmetic.  This is
    what lets the decode loop choose its group size freely."""
    r = rng(91 + w8)
    K, N, ksplit = 1280, 1280, 4
    A = (r.standard_normal((256, K)) * 0.5).astype(np.float16)
    Wf = (r.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    if w8 == 4:
        q, s = symmetric_quantize_int4(Wf); tiles = W.tile_linear_int4(q)
    elif w8:
        q, s = symmetric_quantize_int8(Wf); tiles = W.tile_linear(q)
    else:
        s, tiles = None, W.tile_linear(Wf)
    t_dev, s_dev = dev(tiles.view(np.uint8)), (dev(s) if s is not None else None)
    a_dev = dev(A)
    def run(M):
        part = torch.zeros((ksplit, M, N), dtype=torch.float32, device="cuda")
        native.check(lib.wm_gemm_skinny(a_dev.data_ptr(), K, M, K, t_dev.data_ptr(), N // 16, w8,
                                        s_dev.data_ptr() if s_dev is not None else None, ksplit, part.data_ptr(), stream()))
        torch.cuda.synchronize()
        return part
    full = run(256)
    for M in (1, 3, 40, 100, 130, 200):
        assert torch.equal(run(M), full[:, :M]), M


@pytest.mark.parametrize("M,w8,K,N", [(1, 1, 1280, 1280), (3, 1, 1280, 3840), (8, 0, 1280, 1280), (16, 4, 1280, 1280), (20, 1, 5120, 1280),
                                        (32, 0, 1280, 1280), (1, 1, 5120, 1280), (32, 4, 5120, 1280)])
def test_gemv_fused_small_batch_path(lib, M, w8, K, N):
    """wm_gemv_fused (csrc/gemv_small.hip), the one-launch Linear of the small-batch decode path, mode by mode, against
    fp32 restatements with the reference's rounding points (fp16 Linear output, then the element-wise op in fp32, rounded):
      * mode 0: the fp32 sums against numpy and against the slab form (wm_gemm_skinny, slabs added): same products, another
        order of the K slices -> a few fp32 ulps;
      * the kernel's own LayerNorm of the input rows against torch's fp32 LayerNorm of the same fp16 rows;
      * mode 1 (bias + erf GELU), mode 2 (bias + residual in place: exact), mode 3 (logits, ragged vocabulary edge);
      * a row's result does not depend on how many rows share the launch (bit for bit)."""
    r = rng(700 + M + w8 + K)
    A = (r.standard_normal((M, K)) * 0.7 + 0.2).astype(np.float16)
    Wf = (r.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    bias = (r.standard_normal(N) * 0.1).astype(np.float16)
    if w8 == 4:
        q, sc = symmetric_quantize_int4(Wf); tiles = W.tile_linear_int4(q)
        Wd = q.astype(np.float32) * sc.astype(np.float32)[:, None]
    elif w8:
        q, sc = symmetric_quantize_int8(Wf); tiles = W.tile_linear(q)
        Wd = q.astype(np.float32) * sc.astype(np.float32)[:, None]
    else:
        sc, tiles, Wd = None, W.tile_linear(Wf), Wf.astype(np.float32)
    t_dev, s_dev, a_dev, b_dev = dev(tiles.view(np.uint8)), (dev(sc) if sc is not None else None), dev(A), dev(bias)

    def call(mode, a, m=M, **kw):
        io = native.WmGemvIO()
        io.a, io.lda, io.m, io.k = a.data_ptr(), K, m, K
        io.wt, io.n_blocks, io.w8 = t_dev.data_ptr(), N // 16, w8
        io.scale = s_dev.data_ptr() if s_dev is not None else None
        io.mode, io.bias, io.gelu_kind = mode, b_dev.data_ptr(), 1
        for k_, v in kw.items():
            setattr(io, k_, v)
        native.check(lib.wm_gemv_fused(C.byref(io), stream()), "wm_gemv_fused")
        torch.cuda.synchronize()

    # ---- mode 0
    out32 = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    call(0, a_dev, out32=out32.data_ptr(), ld32=N)
    want = A.astype(np.float32) @ Wd.T
    scale_ = max(1.0, np.abs(want).max())
    assert np.abs(out32.cpu().numpy() - want).max() < 2e-5 * scale_ * np.sqrt(K / 1280)
    ks = lib.wm_gemm_skinny_default_ksplit(M, K, N // 16, w8)
    part = torch.zeros((ks, M, N), dtype=torch.float32, device="cuda")
    native.check(lib.wm_gemm_skinny(a_dev.data_ptr(), K, M, K, t_dev.data_ptr(), N // 16, w8,
                                    s_dev.data_ptr() if s_dev is not None else None, ks, part.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert float((out32 - part.sum(0)).abs().max()) < 2e-5 * scale_ * np.sqrt(K / 1280)
    if M > 1:                                                  # batch independence: row 0 alone == row 0 of the batch
        solo = torch.zeros((1, N), dtype=torch.float32, device="cuda")
        call(0, a_dev, m=1, out32=solo.data_ptr(), ld32=N)
        assert torch.equal(solo[0], out32[0])
    y16 = (out32.cpu().numpy() + bias.astype(np.float32)).astype(np.float16).astype(np.float32)     # the Linear's fp16 output

    # ---- mode 1: GELU
    h = torch.zeros((M, N), dtype=torch.float16, device="cuda")
    call(1, a_dev, out16=h.data_ptr(), ld16=N, n_valid=N)
    ref = torch.nn.functional.gelu(torch.from_numpy(y16)).half().float().numpy()
    assert np.abs(h.float().cpu().numpy() - ref).max() <= 2.0 ** -10 * max(1.0, np.abs(ref).max())

    # ---- mode 2: residual stream in place
    x0 = (r.standard_normal((M, N)) * 1.5).astype(np.float16)
    x = dev(x0)
    call(2, a_dev, x=x.data_ptr(), ldx=N)
    x_ref = (x0.astype(np.float32) + y16).astype(np.float16)
    assert np.array_equal(x.cpu().numpy(), x_ref)                                   # rounding points are the reference's: exact

    # ---- mode 3: logits with a ragged vocabulary edge
    lg = torch.full((M, N), 7.0, dtype=torch.float16, device="cuda")
    call(3, a_dev, out16=lg.data_ptr(), ld16=N, n_valid=N - 5)
    assert torch.equal(lg[:, :N - 5], out32.half()[:, :N - 5]) and bool((lg[:, N - 5:] == 7.0).all())

    # ---- LayerNorm of the input rows inside the kernel (the LayerNorm'ed inputs are residual-stream rows: K = 1280)
    if K == 1280:
        gam = (1 + r.uniform(-0.1, 0.1, K)).astype(np.float16)
        bet = r.uniform(-0.1, 0.1, K).astype(np.float16)
        g_dev, be_dev = dev(gam), dev(bet)
        out_ln = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        call(0, a_dev, out32=out_ln.data_ptr(), ld32=N, ln_gamma=g_dev.data_ptr(), ln_beta=be_dev.data_ptr())
        xn = torch.nn.functional.layer_norm(torch.from_numpy(A.astype(np.float32)), (K,), torch.from_numpy(gam.astype(np.float32)),
                                            torch.from_numpy(bet.astype(np.float32)), 1e-5).half()
        out_pre = torch.zeros((M, N), dtype=torch.float32, device="cuda")             # the same GEMV on torch's LayerNorm output
        call(0, xn.cuda(), out32=out_pre.data_ptr(), ld32=N)
        want_ln = xn.float().numpy() @ Wd.T
        # two-pass fp32 statistics like torch's: the normalised rows differ in a handful of last fp16 bits at most
        assert float((out_pre - out_ln).abs().max()) < 1e-3 * max(1.0, np.abs(want_ln).max())
        assert np.abs(out_ln.cpu().numpy() - want_ln).max() < 2e-3 * max(1.0, np.abs(want_ln).max())


def test_attn_decode_cross_rows_do_not_depend_on_the_launch(lib):
    """One workgroup per (utterance, head) for 12 utterances, the persistent balanced launch for 64: same bits per row."""
    r = rng(123)
    H, Tk = 20, 1500
    q = r.standard_normal((64, H * 64)).astype(np.float32)
    kv = torch.from_numpy(r.standard_normal((64, 2, H, Tk, 64)).astype(np.float16)).cuda()
    qd = dev(q)
    def run(B):
        out = torch.zeros((B, H * 64), dtype=torch.float16, device="cuda")
        native.check(lib.wm_attn_decode_cross(qd.data_ptr(), B, 1, H, Tk, kv.data_ptr(), out.data_ptr(), 1, None, stream()))
        torch.cuda.synchronize()
        return out
    big = run(64)                                                     # 1280 items >= 4 per CU: persistent
    assert torch.equal(run(12), big[:12])                             # 240 items: one workgroup each
    assert len({tuple(row) for row in big[:4].float().cpu().numpy().round(3).tolist()}) > 1


def test_attn_decode_cross_split_rows_do_not_depend_on_the_launch(lib):
    """Key range cut into 8 splits of 188 keys, merged by the merge kernel: 1 or 2 utterances (160 / 320 items, one workgroup
    each) against 8 utterances in the persistent launch (1280 items): same bits per row."""
    r = rng(321)
    H, Tk, ns = 20, 1500, 8
    q = r.standard_normal((8, H * 64)).astype(np.float32)
    kv = torch.from_numpy(r.standard_normal((8, 2, H, Tk, 64)).astype(np.float16)).cuda()
    qd = dev(q)
    def run(B):
        out = torch.zeros((B, H * 64), dtype=torch.float16, device="cuda")
        ws = torch.zeros(B * H * ns * 66, dtype=torch.float32, device="cuda")
        native.check(lib.wm_attn_decode_cross(qd.data_ptr(), B, 1, H, Tk, kv.data_ptr(), out.data_ptr(), ns, ws.data_ptr(), stream()))
        torch.cuda.synchronize()
        return out
    big = run(8)
    assert torch.equal(run(2), big[:2]) and torch.equal(run(1), big[:1])
    assert bool((big != 0).any())


@pytest.mark.parametrize("M,N,K", [(1, 1024, 4096), (64, 1536, 4096)])
def test_weight_only_matmul_reference_spec(lib, M, N, K):
    """The reference's own known-answer test (test_weight_only_quant_matmul.py:94-119): uniform
    weights, activations * 200, oracle = fp32 (x @ q) * scale -> fp16, tolerance 1.5*colmax/128."""
    g = torch.Generator().manual_seed(0)
    x = ((torch.rand((M, K), generator=g) * 2 - 1) * 200.0).half().numpy()
    w = (torch.rand((K, N), generator=g) * 2 - 1).half().numpy()          # [K, N] like the reference
    got, _ = _run_skinny(lib, x, np.ascontiguousarray(w.T), 1, lib.wm_gemm_skinny_default_ksplit(M, K, N // 16, 1))
    q, s = symmetric_quantize_int8(np.ascontiguousarray(w.T))
    ref = woq_reference_matmul(x, q.T, s).astype(np.float32)
    atol = woq_colwise_atol(ref)
    assert (np.abs(got.astype(np.float16).astype(np.float32) - ref) <= atol[None, :] + 1e-2).all()
    # and far tighter than the reference's bound: we only differ by the fp16 store
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.abs(ref).max()


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,N", [(4, 128), (7, 384), (33, 1280), (3, 5120)])
def test_layernorm(lib, M, N):
    r = rng(N)
    x = (r.standard_normal((M, N)) * 3 + 1).astype(np.float16)
    g = r.uniform(0.5, 1.5, N).astype(np.float16)
    b = r.uniform(-0.5, 0.5, N).astype(np.float16)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x).float(), (N,), torch.from_numpy(g).float(),
                                         torch.from_numpy(b).float(), 1e-5).numpy()
    xd, gd, bd = dev(x), dev(g), dev(b)
    out = torch.empty((M, N), dtype=torch.float16, device="cuda")
    native.check(lib.wm_layernorm(xd.data_ptr(), N, M, N, gd.data_ptr(), bd.data_ptr(), out.data_ptr(), N, stream()))
    torch.cuda.synchronize()
    # reference accepts 2e-2 for fp16 (test_layer_norm.py:74-78); we are at one fp16 ulp
    assert np.abs(out.float().cpu().numpy() - ref).max() <= 2.0 ** -10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,N", [(1, 1280), (7, 1280), (4501, 1280), (33, 384), (9, 128), (5, 1536), (10, 1024)])
def test_layernorm_streaming_form_is_bit_identical(lib, M, N, monkeypatch):
    """The streaming LayerNorm (one wave per row, rows in registers: the encoder's 65 launches over 1500 x batch rows) adds
    up a row in the order of the one-workgroup-per-row form inside row_finish_kernel, so the two are interchangeable bit
    for bit -- ragged row counts (the last wave's second row, the last workgroup's idle waves) and every row width included."""
    r = rng(M + N)
    x = dev((r.standard_normal((M, N)) * 3 + 1).astype(np.float16))
    g, b = dev(r.uniform(0.5, 1.5, N).astype(np.float16)), dev(r.uniform(-0.5, 0.5, N).astype(np.float16))
    outs = []
    monkeypatch.setenv("WM_LAB", "1")                         # lab knobs are honoured only under WM_LAB=1
    for form in ("", "workgroup"):
        monkeypatch.setenv("WM_LN_FORM", form)
        out = torch.full((M + 1, N), 7.0, dtype=torch.float16, device="cuda")
        native.check(lib.wm_layernorm(x.data_ptr(), N, M, N, g.data_ptr(), b.data_ptr(), out.data_ptr(), N, stream()))
        torch.cuda.synchronize()
        assert bool((out[M] == 7.0).all())                    # nothing written past the last row
        outs.append(out[:M].clone())
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------ encoder attention
@pytest.mark.parametrize("B,T,H", [(1, 64, 1), (2, 150, 2), (1, 1500, 3), (3, 300, 2)])
def test_attn_encoder(lib, B, T, H):
    r = rng(T + H)
    C_ = H * 64
    q = (r.standard_normal((B, T, C_))).astype(np.float16)
    k = (r.standard_normal((B, T, C_))).astype(np.float16)
    v = (r.standard_normal((B, T, C_))).astype(np.float16)
    m = OracleModel(Dims(80, T, C_, H, 0, 8, 4, C_, H, 0), {}, OracleConfig(act="float16"))
    ref = m._attend(torch.from_numpy(q).float(), torch.from_numpy(k).float(), torch.from_numpy(v).float(), H).numpy()
    scale = 64 ** -0.25
    qs = (q.astype(np.float32) * scale).astype(np.float16)        # what the QKV GEMM epilogue hands over
    ks = (k.astype(np.float32) * scale).astype(np.float16)
    qkv = dev(np.concatenate([qs, ks, v], axis=2).reshape(B * T, 3 * C_))
    out = torch.zeros((B * T, C_), dtype=torch.float16, device="cuda")
    native.check(lib.wm_attn_encoder(qkv.data_ptr(), 3 * C_, B, T, H, out.data_ptr(), C_, stream()))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy().reshape(B, T, C_)
    # online softmax with fp16 probabilities vs the oracle's normalised fp16 probabilities:
    # both carry 2^-11 relative rounding per probability -> 2e-3 absolute on O(1) outputs
    assert np.abs(got - ref).max() <= 2e-3, np.abs(got - ref).max()


# -------------------------------------------------------------------------------- decode cross-attention
@pytest.mark.parametrize("B,L,H,Tk,nsplit", [(2, 1, 2, 100, 1), (2, 3, 2, 100, 1), (1, 1, 20, 1500, 1),
                                             (3, 1, 2, 1500, 4), (2, 3, 3, 333, 3), (1, 2, 1, 8, 1),
                                             # single-token items of one or two blocks of rows (<= 256 keys per split)
                                             (1, 1, 20, 1500, 8), (2, 1, 2, 250, 1), (1, 1, 3, 129, 1), (4, 1, 2, 1500, 7), (2, 1, 2, 257, 1)])
def test_attn_decode_cross(lib, B, L, H, Tk, nsplit):
    r = rng(B + L + H + Tk)
    C_ = H * 64
    q = r.standard_normal((B * L, C_)).astype(np.float16).astype(np.float32)
    kv = r.standard_normal((B, 2, H, Tk, 64)).astype(np.float16)
    m = OracleModel(Dims(80, Tk, C_, H, 0, 8, 4, C_, H, 0), {}, OracleConfig(act="float16"))
    kk = torch.from_numpy(kv[:, 0]).float().permute(0, 2, 1, 3).reshape(B, Tk, C_)
    vv = torch.from_numpy(kv[:, 1]).float().permute(0, 2, 1, 3).reshape(B, Tk, C_)
    ref = m._attend(torch.from_numpy(q).reshape(B, L, C_), kk, vv, H).numpy().reshape(B * L, C_)
    qd, kvd = dev(q), dev(kv)
    out = torch.zeros((B * L, C_), dtype=torch.float16, device="cuda")
    ws = torch.zeros(B * H * nsplit * L * 66, dtype=torch.float32, device="cuda")
    native.check(lib.wm_attn_decode_cross(qd.data_ptr(), B, L, H, Tk, kvd.data_ptr(), out.data_ptr(), nsplit,
                                          ws.data_ptr(), stream()))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    tol = 1e-3 if nsplit == 1 else 2e-3        # single pass rounds exactly where the oracle does
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()


@pytest.mark.parametrize("B,L,H,Tk,nsplit", [(2, 1, 2, 100, 1), (2, 3, 2, 100, 1), (1, 1, 20, 1500, 1), (3, 1, 2, 1500, 4),
                                             (2, 4, 3, 333, 3), (1, 2, 1, 8, 1), (130, 1, 20, 64, 1)])
def test_attn_decode_cross_int8(lib, B, L, H, Tk, nsplit):
    """Opt-in int8 cross K/V (beyond the reference): codes = sat_s8(rne(x / t)), attention uses code * t exactly; against the
    oracle's attention over those values.  (130 utterances x 20 heads takes the persistent-launch path.)"""
    r = rng(7 * B + L + H + Tk)
    C_ = H * 64
    q = r.standard_normal((B * L, C_)).astype(np.float16).astype(np.float32)
    kv = r.standard_normal((B, 2, H, Tk, 64)).astype(np.float16)
    t = float(np.float32(np.abs(kv).max()) / np.float32(127.0))
    codes = kv_quantize(torch.from_numpy(kv), t)
    deq = codes.float() * float(np.float32(t))                       # the mode's values are exactly code * t
    m = OracleModel(Dims(80, Tk, C_, H, 0, 8, 4, C_, H, 0), {}, OracleConfig(act="float16"))
    kk = deq[:, 0].permute(0, 2, 1, 3).reshape(B, Tk, C_)
    vv = deq[:, 1].permute(0, 2, 1, 3).reshape(B, Tk, C_)
    ref = m._attend(torch.from_numpy(q).reshape(B, L, C_), kk, vv, H, k_exact=True).numpy().reshape(B * L, C_)
    qd, cd = dev(q), codes.cuda()
    out = torch.zeros((B * L, C_), dtype=torch.float16, device="cuda")
    ws = torch.zeros(B * H * nsplit * L * 66, dtype=torch.float32, device="cuda")
    native.check(lib.wm_attn_decode_cross_i8(qd.data_ptr(), B, L, H, Tk, cd.data_ptr(), t, out.data_ptr(), nsplit,
                                             ws.data_ptr(), stream()))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    tol = 1e-3 if nsplit == 1 else 2e-3
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()


# --------------------------------------------------------------------------------- decode self-attention
@pytest.mark.parametrize("waves", [1, 4])       # one wave per (utterance, head) | the four-wave workgroup form of small groups
@pytest.mark.parametrize("int8_kv", [0, 1])
@pytest.mark.parametrize("B,L,T,H,inplace", [(2, 3, 0, 2, True), (2, 1, 5, 2, True), (1, 1, 130, 3, False),
                                             (3, 1, 447, 2, True), (2, 3, 7, 1, False), (2, 1, 64, 2, True),
                                             (1, 4, 257, 2, True), (2, 2, 300, 1, False)])
def test_attn_decode_self(lib, B, L, T, H, inplace, int8_kv, waves):
    r = rng(B + L + T + H + int8_kv)
    C_ = H * 64
    cap = 448 if inplace else T + L
    t_scale = 0.031
    qkv = r.standard_normal((B * L, 3 * C_)).astype(np.float16).astype(np.float32)
    past_f = (r.standard_normal((B, 2, H, T, 64)) * 1.2).astype(np.float16)
    m = OracleModel(Dims(80, 8, C_, H, 0, 8, 512, C_, H, 0), {}, OracleConfig(act="float16"))
    qkv_t = torch.from_numpy(qkv).reshape(B, L, 3, H, 64)
    new = qkv_t[:, :, 1:].permute(0, 2, 3, 1, 4)                       # [B, 2, H, L, 64]
    if int8_kv:
        past_q = kv_quantize(torch.from_numpy(past_f).float(), t_scale)
        full = torch.cat([kv_dequantize(past_q, t_scale, "float16"), new], dim=3)
        want_present = torch.cat([past_q, kv_quantize(new, t_scale)], dim=3)
        past_store = past_q
    else:
        full = torch.cat([torch.from_numpy(past_f).float(), new], dim=3)
        want_present = full.half()
        past_store = torch.from_numpy(past_f)
    mask = torch.zeros(L, T + L)
    mask[:, T:] = torch.full((L, L), float("-inf")).triu_(1)
    k_all = full[:, 0].permute(0, 2, 1, 3).reshape(B, T + L, C_)
    v_all = full[:, 1].permute(0, 2, 1, 3).reshape(B, T + L, C_)
    ref = m._attend(qkv_t[:, :, 0].reshape(B, L, C_), k_all, v_all, H, mask).numpy().reshape(B * L, C_)

    dt = torch.int8 if int8_kv else torch.float16
    present = torch.zeros((B, 2, H, cap, 64), dtype=dt, device="cuda")
    if inplace:
        present[:, :, :, :T] = past_store.cuda()
        past, past_cap = present, cap
    else:
        past, past_cap = (past_store.cuda().contiguous() if T > 0 else None), T
    out = torch.zeros((B * L, C_), dtype=torch.float16, device="cuda")
    qd = dev(qkv)
    prev = lib.wm_set_self_attn_waves(waves)
    try:
        native.check(lib.wm_attn_decode_self(qd.data_ptr(), B, L, T, H, past.data_ptr() if past is not None else None,
                                             past_cap, present.data_ptr(), cap, int8_kv, t_scale, out.data_ptr(), stream()))
        torch.cuda.synchronize()
    finally:
        lib.wm_set_self_attn_waves(prev)
    got = out.float().cpu().numpy()
    # reference accepts 5e-3 with int8 KV (test_gpt_attention.py:684-693)
    assert np.abs(got - ref).max() <= 1.5e-3, np.abs(got - ref).max()
    got_present = present[:, :, :, :T + L].cpu()
    if int8_kv:
        assert torch.equal(got_present, want_present)                  # integer work: bit-exact
    else:
        assert torch.equal(got_present, want_present)


def test_quantize_i8_exact(lib):
    x = torch.randn(100003, generator=torch.Generator().manual_seed(0)).half()
    x[:6] = torch.tensor([0.5, 1.5, 2.5, -0.5, 1000.0, -1000.0]).half()
    for inv in (1.0, 2.5, 1.0 / 0.4, 33.3):
        q = torch.zeros(x.numel(), dtype=torch.int8, device="cuda")
        xd = x.cuda()
        native.check(lib.wm_quantize_i8(xd.data_ptr(), q.data_ptr(), x.numel(), inv, stream()))
        torch.cuda.synchronize()
        want = torch.clamp(torch.round(x.float() * float(np.float32(inv))), -128, 127).to(torch.int8)
        assert torch.equal(q.cpu(), want)


# ------------------------------------------------------------------------------------------- greedy step
def test_greedy_step_matches_reference_rules(lib, golden_dir):
    fixr = np.load(os.path.join(golden_dir, "decoding_rules.npz"))
    ids = DR.MULTILINGUAL
    V = ids.n_vocab
    sup = sorted(set(fixr["suppress"].tolist() + [ids.no_timestamps]))
    sup_d, blank_d = dev(np.array(sup, dtype=np.int32)), dev(fixr["blank"].astype(np.int32))
    for c, (toks, logits) in enumerate(DR.golden_rule_cases()):
        cur = len(toks)
        tok_buf = torch.zeros((1, 64), dtype=torch.int32, device="cuda")
        tok_buf[0, :cur] = torch.from_numpy(toks).int().cuda()
        lg = dev(logits.astype(np.float16))
        s = torch.zeros(1, dtype=torch.float32, device="cuda")
        n_done = torch.zeros(1, dtype=torch.int32, device="cuda")
        io = native.WmGreedyIO()
        io.logits, io.row_stride, io.batch, io.n_vocab = lg.data_ptr(), V, 1, V
        io.tokens, io.tokens_ld, io.cur_len = tok_buf.data_ptr(), 64, cur
        io.sum_logprobs, io.suppress, io.n_suppress = s.data_ptr(), sup_d.data_ptr(), len(sup)
        io.blank, io.n_blank = blank_d.data_ptr(), len(fixr["blank"])
        io.sample_begin, io.eot, io.timestamp_begin = 3, ids.eot, ids.timestamp_begin
        io.max_initial_timestamp_index, io.apply_rules, io.n_done = 50, 1, n_done.data_ptr()
        native.check(lib.wm_greedy_step(C.byref(io), stream()))
        torch.cuda.synchronize()
        assert int(tok_buf[0, cur]) == int(fixr[f"c{c}_next"]), f"case {c}"
        assert abs(float(s[0]) - float(fixr[f"c{c}_sumlp"])) < 2e-4, f"case {c}"
        assert bool(int(n_done[0])) == bool(fixr[f"c{c}_done"]), f"case {c}"


def test_greedy_step_long_histories_match_oracle_rules(lib, golden_dir):
    """The greedy kernel looks at the sampled tokens with one thread per token (last / penultimate token, latest timestamp):
    histories that span several waves and several passes of the workgroup, timestamps at every kind of position, several rows
    per launch -- against the oracle's restatement of the rules (pinned to the reference's classes by decoding_rules.npz)."""
    fixr = np.load(os.path.join(golden_dir, "decoding_rules.npz"))
    ids = DR.MULTILINGUAL
    V, tb = ids.n_vocab, ids.timestamp_begin
    sup = sorted(set(fixr["suppress"].tolist() + [ids.no_timestamps]))
    rules = DR.RuleSet(ids, 3, sup, fixr["blank"].tolist(), 50)
    sup_d, blank_d = dev(np.array(sup, dtype=np.int32)), dev(fixr["blank"].astype(np.int32))
    r = rng(2026)
    ld = 1500                                              # room for histories longer than one pass of 1024 threads
    for n in (1, 2, 63, 64, 65, 129, 300, 1100):
        rows = []
        for kind in range(6):
            hist = r.integers(0, 50000, n)
            if kind >= 1:                                  # a non-decreasing run of timestamps somewhere
                pos = sorted(r.choice(n, size=min(n, 1 + kind), replace=False).tolist())
                t = int(r.integers(0, 700))
                for q in pos:
                    t += int(r.integers(0, 40)); hist[q] = tb + t
            if kind == 2: hist[-1] = tb + 1200             # ends on a timestamp
            if kind == 3 and n >= 2: hist[-1] = hist[-2] = tb + 1300          # ends on a pair
            if kind == 4: hist[0] = tb + 7; hist[1:] = r.integers(0, 50000, n - 1)   # the only timestamp is the oldest token
            if kind == 5 and n > 70: hist[63] = tb + 900; hist[64:] = r.integers(0, 50000, n - 64)   # last lane of the first wave
            rows.append(np.concatenate([[ids.sot, ids.lang0, ids.transcribe], hist]).astype(np.int64))
        B, cur = len(rows), 3 + n
        toks = np.stack(rows)
        logits = (r.standard_normal((B, V)) * 2.0).astype(np.float32)
        for b in range(B):                                 # clear winners in both classes: no near-ties of the rules
            logits[b, int(r.integers(0, 50000))] += 12.0
            logits[b, tb + int(r.integers(0, 1501))] += 9.0 if b % 2 else 14.0
        logits = logits.astype(np.float16).astype(np.float32)
        dom = []
        filtered = DR.apply_filters(logits, toks, rules, dominance_out=dom)
        want_sum = np.zeros(B, dtype=np.float32)
        want_toks, _ = DR.greedy_update(toks, filtered, want_sum, ids.eot)
        tok_buf = torch.zeros((B, ld), dtype=torch.int32, device="cuda")
        tok_buf[:, :cur] = torch.from_numpy(toks).int().cuda()
        lg = dev(logits.astype(np.float16))
        s = torch.zeros(B, dtype=torch.float32, device="cuda")
        n_done = torch.zeros(1, dtype=torch.int32, device="cuda")
        io = native.WmGreedyIO()
        io.logits, io.row_stride, io.batch, io.n_vocab = lg.data_ptr(), V, B, V
        io.tokens, io.tokens_ld, io.cur_len = tok_buf.data_ptr(), ld, cur
        io.sum_logprobs, io.suppress, io.n_suppress = s.data_ptr(), sup_d.data_ptr(), len(sup)
        io.blank, io.n_blank = blank_d.data_ptr(), len(fixr["blank"])
        io.sample_begin, io.eot, io.timestamp_begin = 3, ids.eot, tb
        io.max_initial_timestamp_index, io.apply_rules, io.n_done = 50, 1, n_done.data_ptr()
        native.check(lib.wm_greedy_step(C.byref(io), stream()))
        torch.cuda.synchronize()
        got = tok_buf[:, cur].cpu().numpy()
        for b in range(B):
            assert abs(dom[b]) > 0.05, (n, b, dom[b])      # the case is not a near-tie of the timestamp-dominance rule
            assert int(got[b]) == int(want_toks[b, -1]), (n, b, int(got[b]), int(want_toks[b, -1]))
            assert abs(float(s[b]) - float(want_sum[b])) < 2e-4, (n, b)


@pytest.mark.gpu
def test_log_mel_device(lib, golden_dir):
    """wm_log_mel (device STFT + mel) against the reference's log_mel_spectrogram output (tests/golden/mel.npz)
    and, at the full 30 s size, against the torch.stft mirror clip by clip (silence, padding tail, loud clip)."""
    import whisper_utils as wu
    g = np.load(os.path.join(golden_dir, "mel.npz"))
    rng = np.random.Generator(np.random.Philox(int(g["audio_seed"])))
    audio = (rng.standard_normal(int(g["n_audio"])) * 0.1).astype(np.float32)
    padded = torch.from_numpy(wu.pad_or_trim(audio, int(g["n_padded"]))).cuda()
    MEL_ATOL = 5e-4        # fp32 direct DFT vs pocketfft, after log10 and /4
    mel = wu.log_mel_spectrogram_device(padded, dtype=torch.float32)
    assert tuple(mel.shape) == g["mel"].shape
    np.testing.assert_allclose(mel.cpu().numpy(), g["mel"], atol=MEL_ATOL)
    mel16 = wu.log_mel_spectrogram_device(padded)
    assert mel16.dtype == torch.float16 and torch.equal(mel16, mel.half())

    gen = torch.Generator().manual_seed(77)
    clips = torch.zeros(4, wu.N_SAMPLES)
    clips[0] = torch.randn(wu.N_SAMPLES, generator=gen) * 0.05
    clips[1, :200000] = torch.sin(torch.arange(200000) * 0.05) * 0.5 + torch.randn(200000, generator=gen) * 0.01
    clips[3] = torch.randn(wu.N_SAMPLES, generator=gen).clamp(-1, 1)
    dev = wu.log_mel_spectrogram_device(clips.cuda(), dtype=torch.float32).cpu()
    assert tuple(dev.shape) == (4, 80, wu.N_FRAMES)
    for b in range(4):
        ref = wu.log_mel_spectrogram(clips[b])
        np.testing.assert_allclose(dev[b].numpy(), ref.numpy(), atol=MEL_ATOL, err_msg=f"clip {b}")
    assert float((dev[2] + 1.5).abs().max()) < 1e-6           # silence: log10(1e-10) everywhere
    # ragged batch view: a strided slice of a bigger buffer (audio_ld > n_samples)
    big = torch.zeros(2, wu.N_SAMPLES + 320, device="cuda")
    big[:, :wu.N_SAMPLES] = clips[:2].cuda()
    view = big[:, :wu.N_SAMPLES]
    assert not view.is_contiguous()
    out_view = wu.log_mel_spectrogram_device(view.contiguous(), dtype=torch.float32).cpu()
    assert torch.equal(out_view, dev[:2])
