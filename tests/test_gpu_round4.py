"""Round 4 GPU tests (through the C ABI, on a real MI355X)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import native  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return native.load_library()


def stream():
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------ GEMM tile forms
@pytest.mark.parametrize("M,N,K,act,resid", [
    (1500, 1280, 1280, 0, True), (3000, 3840, 1280, 0, False), (700, 5120, 1280, 1, False), (1500, 1280, 5120, 0, True),
    (131, 256, 128, 1, True), (6000, 2560, 1280, 0, False),
])
def test_gemm_small_tiles_bit_identical(lib, M, N, K, act, resid):
    """The 128 x 128 form (few rows: one to a few clips) and the persistent 256 x 256 kernel give the same bits: an output element
    is the same chain of MFMA steps over K and the same epilogue arithmetic in both -- which is what lets the tile follow the
    size of the launch without a clip's encoder output depending on its batch (R/tensorrt_llm/models/whisper/model.py:149-172
    is batch-independent by construction)."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).half()
    bias = (torch.randn(N, device="cuda", generator=g) * 0.1).half()
    R = (torch.randn(M, N, device="cuda", generator=g) * 0.5).half() if resid else None
    outs = []
    prev = lib.wm_set_gemm_small_tiles(-1)
    try:
        for tiles in (0, 1 << 30):
            lib.wm_set_gemm_small_tiles(tiles)
            C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
            native.check(lib.wm_gemm(A.data_ptr(), K, M, K, W.data_ptr(), N, 0, None, bias.data_ptr(),
                                     R.data_ptr() if resid else None, N, act, C.data_ptr(), N, None, 0, stream()))
            torch.cuda.synchronize()
            outs.append(C)
    finally:
        lib.wm_set_gemm_small_tiles(prev)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    ref = (A.float() @ W.float().T + bias.float()).half().float()
    if act == 1:
        ref = torch.nn.functional.gelu(ref).half().float()
    if resid:
        ref = (ref + R.float()).half().float()
    tol = 2.0 ** -10 * max(1.0, float(ref.abs().max()))          # one fp16 ulp of the output magnitude
    assert float((outs[1].float() - ref).abs().max()) <= tol


def test_gemm_tile_switch_default_and_setter(lib):
    prev = lib.wm_set_gemm_small_tiles(7)
    assert lib.wm_set_gemm_small_tiles(-1) == 7
    assert lib.wm_set_gemm_small_tiles(prev) == 192
