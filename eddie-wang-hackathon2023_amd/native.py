"""ctypes binding of libwhisper_mi355.so (include/whisper_mi355.h).

The counterpart of the reference's plugin loader (R/tensorrt_llm/plugin/plugin.py:10-22:
`ctypes.CDLL(libnvinfer_plugin_tensorrt_llm.so)` + `initLibNvInferPlugins`).  There is NO
fallback: if the HIP library is missing or fails to load, importing the engine raises.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
from typing import Optional, Sequence

# Decode steps are replayed as captured hipGraphs of ~300 dependent short kernels.  ROCm 7's default replays a graph from
# pre-built AQL packets (DEBUG_CLR_GRAPH_PACKET_CAPTURE=1); with the nodes enqueued one by one through the ordinary dispatch
# path instead, every kernel of the chain costs ~0.4 us less on the GPU side: token step B = 1 1.685 -> 1.570 ms, B = 8
# 2.262 -> 2.126, B = 32 3.504 -> 3.423, B = 576 unchanged (23.2 ms, 15.5-15.6 k tokens/s either way) --
# profiles/r3ah_bench_*.json, sweep of the other launch-path knobs in profiles/r3ag_launch_knobs.txt.  The flag is read when
# the HIP runtime initialises (first HIP call of the process), so it is set here, at import, unless the caller chose a value:
# import this package before the first torch.cuda call, or export the variable.  Same kernels, same results.
#
# DEBUG_CLR_GRAPH_PACKET_CAPTURE is a debug switch of the HIP runtime, not a documented interface: it is process-wide (every HIP
# user of the process replays its graphs node by node), it takes effect only if it is in the environment before the process's
# first HIP call, and a ROCm update may drop it.  So: a value chosen by the caller wins; WM_GRAPH_NODE_REPLAY=0 keeps this
# package from touching the variable at all; the package records what it did and whether it was in time (RUNTIME_KNOBS,
# runtime_report() -- bench.py and summarize.py print it) and warns when the runtime was already initialised; and
# tests/test_gpu_round4.py::test_graph_node_replay_still_pays fails when the switch stops making the batch-1 step faster.
import sys
import warnings


def _hip_already_initialised() -> bool:
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:       # noqa: BLE001
        return False


def _configure_runtime() -> dict:
    name = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
    late = _hip_already_initialised()
    if name in os.environ:
        return {"name": name, "value": os.environ[name], "set_by": "caller", "in_time": None}      # the caller's business, and timing
    if os.environ.get("WM_GRAPH_NODE_REPLAY", "1") == "0":
        return {"name": name, "value": None, "set_by": "nobody (WM_GRAPH_NODE_REPLAY=0)", "in_time": None}
    os.environ[name] = "0"
    if late:
        warnings.warn("whisper_mi355: the HIP runtime was initialised before this package was imported, so "
                      f"{name}=0 (node-by-node graph replay, 6-10 % of the small-batch token step) is NOT in effect: "
                      "import the package before the first torch.cuda call, or export the variable yourself", RuntimeWarning, stacklevel=3)
    return {"name": name, "value": "0", "set_by": "package", "in_time": not late}


RUNTIME_KNOBS = _configure_runtime()


def runtime_report() -> dict:
    """What this package did to the HIP runtime's environment, whether it was in time, and the lab knobs the library honoured."""
    rep = dict(RUNTIME_KNOBS)
    rep["effective_env"] = os.environ.get(RUNTIME_KNOBS["name"])
    if _lib is not None:
        buf = C.create_string_buffer(4096)
        _lib.wm_lab_knobs(buf, 4096)
        rep["lab_knobs_honoured"] = buf.value.decode() or None
    return rep


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WM_LIBRARY_PATH") or os.path.join(_HERE, "libwhisper_mi355.so")

EXPORTS = (
    "wm_version", "wm_last_error", "wm_device_count", "wm_engine_create", "wm_engine_destroy",
    "wm_engine_info", "wm_engine_weight_bytes", "wm_encoder_workspace_bytes", "wm_encoder_forward",
    "wm_encoder_forward_shared", "wm_encoder_forward_range",
    "wm_cross_kv_workspace_bytes", "wm_cross_kv", "wm_decoder_workspace_bytes", "wm_decoder_step",
    "wm_greedy_step", "wm_gemm", "wm_gemm_skinny", "wm_gemm_skinny_default_ksplit", "wm_layernorm",
    "wm_attn_encoder", "wm_attn_decode_cross", "wm_attn_decode_self", "wm_quantize_i8",
    "wm_profile_configure", "wm_profile_read", "wm_step_advance", "wm_log_mel_workspace_bytes", "wm_log_mel",
    "wm_flac_info", "wm_flac_decode", "wm_conv1d_gelu", "wm_argmax", "wm_gemv_fused", "wm_gemm_rows", "wm_set_rows_path", "wm_set_small_batch_rows", "wm_set_self_attn_waves", "wm_set_gemm_small_tiles", "wm_lab_knobs", "wm_set_cross_v_skip", "wm_set_decode_chain", "wm_decode_chain_error", "wm_decode_chain_status", "wm_debug_occupy", "wm_decoder_step_multi", "wm_stream_create_cu_mask", "wm_stream_destroy", "wm_attn_decode_cross_i8", "wm_debug_timeline",
    "wm_step_finish",
)


ABI_VERSION = 8          # WM_ABI_VERSION of include/whisper_mi355.h this binding was written against

# Held by WhisperDecoding.main_loop for the length of a stream capture and by WhisperEncoding.prefetch's helper thread around every call
# it makes into the HIP runtime (event queries, launches): a runtime call from another thread while a capture is open can invalidate it.
import threading as _threading
CAPTURE_LOCK = _threading.RLock()


class WmError(RuntimeError):
    pass


class WmFlacStreamInfo(C.Structure):
    """wm_flac_streaminfo (include/whisper_mi355.h)."""
    _fields_ = [("sample_rate", C.c_int32), ("channels", C.c_int32), ("bits_per_sample", C.c_int32),
                ("max_block_size", C.c_int32), ("total_samples", C.c_int64), ("md5", C.c_uint8 * 16)]


class WmChainStatus(C.Structure):
    """wm_chain_status (include/whisper_mi355.h)."""
    _fields_ = [("mode", C.c_int32), ("declined", C.c_int32), ("error_pending", C.c_int32), ("pad_", C.c_int32),
                ("launches", C.c_int64), ("declined_calls", C.c_int64), ("reason", C.c_char * 200), ("footprint", C.c_char * 200)]


def chain_status() -> dict:
    """wm_decode_chain_status of the current device as a dict (tests, diagnostics)."""
    st = WmChainStatus()
    check(load_library().wm_decode_chain_status(C.byref(st)), "wm_decode_chain_status")
    return {"mode": st.mode, "declined": bool(st.declined), "error_pending": bool(st.error_pending), "launches": int(st.launches),
            "declined_calls": int(st.declined_calls), "reason": st.reason.decode(), "footprint": st.footprint.decode()}


class WmDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]

    def to_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class WmDecoderIO(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("n_new", C.c_int32), ("n_past", C.c_int32),
        ("tokens", C.c_void_p), ("tokens_ld", C.c_int32), ("positional_embedding", C.c_void_p),
        ("past", C.POINTER(C.c_void_p)), ("past_capacity", C.c_int32),
        ("present", C.POINTER(C.c_void_p)), ("present_capacity", C.c_int32),
        ("cross", C.POINTER(C.c_void_p)),
        ("logits", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("qkv_amax", C.c_void_p),
        ("n_past_dev", C.c_void_p),
        ("live_rows", C.c_void_p),
        ("workspace_id", C.c_uint64),
        ("not_alone", C.c_int32),
    ]


class WmGemvIO(C.Structure):
    """wm_gemv_io (include/whisper_mi355.h)."""
    _fields_ = [
        ("a", C.c_void_p), ("lda", C.c_int32), ("m", C.c_int32), ("k", C.c_int32),
        ("wt", C.c_void_p), ("n_blocks", C.c_int32), ("w8", C.c_int32), ("scale", C.c_void_p),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
        ("mode", C.c_int32), ("bias", C.c_void_p), ("gelu_kind", C.c_int32),
        ("out32", C.c_void_p), ("ld32", C.c_int32),
        ("out16", C.c_void_p), ("ld16", C.c_int32), ("n_valid", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int32),
    ]


class WmGreedyIO(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p), ("row_stride", C.c_int64),
        ("batch", C.c_int32), ("n_vocab", C.c_int32),
        ("tokens", C.c_void_p), ("tokens_ld", C.c_int32), ("cur_len", C.c_int32),
        ("sum_logprobs", C.c_void_p),
        ("suppress", C.c_void_p), ("n_suppress", C.c_int32),
        ("blank", C.c_void_p), ("n_blank", C.c_int32),
        ("sample_begin", C.c_int32), ("eot", C.c_int32), ("timestamp_begin", C.c_int32),
        ("max_initial_timestamp_index", C.c_int32),
        ("apply_rules", C.c_int32),
        ("n_done", C.c_void_p),
        ("n_past_dev", C.c_void_p),
        ("done", C.c_void_p),
        ("row_limit", C.c_void_p),
        ("temperature", C.c_float), ("row0", C.c_int32),
        ("seed", C.c_uint64), ("seed_dev", C.c_void_p),
    ]


_lib: Optional[C.CDLL] = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Load the HIP library, declare prototypes.  Raises WmError when it is absent: the product
    path has no CPU or PyTorch fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise WmError(f"{path} not found: build it with `python __graft_entry__.py` "
                      f"(hipcc --offload-arch=gfx950); there is no fallback path")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise WmError(f"{path} does not export {name}")
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.wm_version.restype = i32
    if lib.wm_version() != ABI_VERSION:       # a stale .so: signatures moved (wm_gemm's workspace), arguments would be misread
        raise WmError(f"{path} has ABI version {lib.wm_version()}, this binding needs {ABI_VERSION}: rebuild it "
                      f"(python __graft_entry__.py)")
    lib.wm_last_error.restype = C.c_char_p
    lib.wm_device_count.argtypes = [C.POINTER(i32)]
    lib.wm_engine_create.argtypes = [vp, sz, i32, C.POINTER(vp)]
    lib.wm_engine_destroy.argtypes = [vp]
    lib.wm_engine_destroy.restype = None
    lib.wm_engine_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(WmDims)]
    lib.wm_engine_weight_bytes.argtypes = [vp]
    lib.wm_engine_weight_bytes.restype = sz
    lib.wm_encoder_workspace_bytes.argtypes = [vp, i32]
    lib.wm_encoder_workspace_bytes.restype = sz
    lib.wm_encoder_forward.argtypes = [vp, vp, i32, vp, vp, sz, vp]
    lib.wm_encoder_forward_shared.argtypes = [vp, vp, i32, vp, vp, sz, i32, vp]
    lib.wm_encoder_forward_range.argtypes = [vp, vp, i32, vp, vp, sz, i32, i32, i32, vp]
    lib.wm_cross_kv_workspace_bytes.argtypes = [vp, i32]
    lib.wm_cross_kv_workspace_bytes.restype = sz
    lib.wm_cross_kv.argtypes = [vp, vp, i32, C.POINTER(vp), vp, sz, vp]
    lib.wm_decoder_workspace_bytes.argtypes = [vp, i32, i32]
    lib.wm_decoder_workspace_bytes.restype = sz
    lib.wm_decoder_step.argtypes = [vp, C.POINTER(WmDecoderIO), vp]
    lib.wm_greedy_step.argtypes = [C.POINTER(WmGreedyIO), vp]
    lib.wm_gemm.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, i32, i32, vp, i32, vp, sz, vp]
    lib.wm_conv1d_gelu.argtypes = [vp, i32, i32, i32, vp, i32, vp, i32, i32, i32, vp, vp]
    lib.wm_argmax.argtypes = [vp, C.c_int64, i32, i32, vp, vp]
    lib.wm_gemv_fused.argtypes = [C.POINTER(WmGemvIO), vp]
    lib.wm_set_small_batch_rows.argtypes = [i32]
    lib.wm_gemm_rows.argtypes = [C.POINTER(WmGemvIO), vp]
    lib.wm_set_rows_path.argtypes = [i32]
    lib.wm_set_self_attn_waves.argtypes = [i32]
    lib.wm_set_gemm_small_tiles.argtypes = [i32]
    lib.wm_lab_knobs.argtypes = [C.c_char_p, sz]
    lib.wm_set_cross_v_skip.argtypes = [i32]
    lib.wm_set_decode_chain.argtypes = [i32]
    lib.wm_decode_chain_error.argtypes = [C.POINTER(C.c_int)]
    lib.wm_decode_chain_status.argtypes = [C.POINTER(WmChainStatus)]
    lib.wm_debug_occupy.argtypes = [i32, sz, C.c_int64, vp]
    lib.wm_gemm_skinny.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, i32, vp, vp]
    lib.wm_gemm_skinny_default_ksplit.argtypes = [i32, i32, i32, i32]
    lib.wm_layernorm.argtypes = [vp, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.wm_attn_encoder.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp]
    lib.wm_attn_decode_cross.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp]
    lib.wm_attn_decode_self.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, i32, i32, C.c_float, vp, vp]
    lib.wm_quantize_i8.argtypes = [vp, vp, C.c_int64, C.c_float, vp]
    lib.wm_log_mel_workspace_bytes.argtypes = [i32, i32, i32]
    lib.wm_log_mel_workspace_bytes.restype = sz
    lib.wm_log_mel.argtypes = [vp, i32, i32, C.c_int64, vp, i32, vp, vp, vp, sz, vp]
    lib.wm_decoder_step_multi.argtypes = [vp, i32, C.POINTER(C.POINTER(WmDecoderIO)), C.POINTER(vp), vp]
    lib.wm_stream_create_cu_mask.argtypes = [C.POINTER(C.c_uint32), i32, C.POINTER(vp)]
    lib.wm_stream_destroy.argtypes = [vp]
    lib.wm_attn_decode_cross_i8.argtypes = [vp, i32, i32, i32, i32, vp, C.c_float, vp, i32, vp, vp]
    lib.wm_flac_info.argtypes = [vp, sz, C.POINTER(WmFlacStreamInfo)]
    lib.wm_flac_decode.argtypes = [vp, sz, vp, C.c_int64, C.POINTER(C.c_int64)]
    lib.wm_step_advance.argtypes = [vp, vp]
    lib.wm_step_finish.argtypes = [vp, vp, i32, vp, vp]
    lib.wm_debug_timeline.argtypes = [vp, i32]
    lib.wm_profile_configure.argtypes = [i32, i32, i32]
    lib.wm_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load_library().wm_last_error().decode(errors="replace")
        raise WmError(f"{what or 'libwhisper_mi355'} failed (rc={rc}): {msg}")


def ptr_array(tensors: Sequence) -> "C.Array":
    """Device pointers of a list of torch tensors (or ints / None) as a void*[]."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else (t.data_ptr() if hasattr(t, "data_ptr") else int(t))
    return arr


class Engine:
    """Owns one wm_engine (device-resident weights of one *.engine file)."""

    def __init__(self, blob: bytes, device: int = 0):
        self.lib = load_library()
        handle = C.c_void_p()
        if not isinstance(blob, bytes):
            blob = bytes(blob)
        # c_char_p points at the bytes object's own buffer: no host copy of a multi-GB blob
        check(self.lib.wm_engine_create(C.cast(C.c_char_p(blob), C.c_void_p), len(blob), device, C.byref(handle)),
              "wm_engine_create")
        self.handle = handle
        kind, flags, dims = C.c_int32(), C.c_uint32(), WmDims()
        check(self.lib.wm_engine_info(self.handle, C.byref(kind), C.byref(flags), C.byref(dims)))
        self.kind, self.flags, self.dims, self.device = kind.value, flags.value, dims.to_dict(), device

    @property
    def weight_bytes(self) -> int:
        return int(self.lib.wm_engine_weight_bytes(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.wm_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_MASKED_STREAMS = {}


def _destroy_masked_streams():
    """Interpreter exit: give the dedicated hardware queues back before the HIP runtime (and any profiler
    attached to it) tears down -- a process that exits with CU-masked queues alive crashes in rocprofv3's finaliser."""
    if not _MASKED_STREAMS or _lib is None:
        return
    try:
        import torch
        torch.cuda.synchronize()
    except Exception:      # noqa: BLE001 -- nothing useful to do at exit
        pass
    for st in _MASKED_STREAMS.values():
        _lib.wm_stream_destroy(st.cuda_stream)
    _MASKED_STREAMS.clear()


atexit.register(_destroy_masked_streams)


def create_masked_stream(cu_enabled, index: int = 0):
    """A torch stream whose kernels run only on the CUs with a true entry in `cu_enabled` (CU i = entry i):
    wm_stream_create_cu_mask wrapped as torch.cuda.ExternalStream (so wait_stream / synchronize work).
    Streams are cached per (device, mask, index) for the life of the process: hardware queues are a scarce
    resource (ROCm multiplexes streams onto GPU_MAX_HW_QUEUES of them), so callers share, never re-create."""
    import torch
    key = (torch.cuda.current_device(), tuple(bool(b) for b in cu_enabled), index)
    if key not in _MASKED_STREAMS:
        n_words = (len(cu_enabled) + 31) // 32
        words = (C.c_uint32 * n_words)()
        for i, on in enumerate(cu_enabled):
            if on:
                words[i // 32] |= 1 << (i % 32)
        handle = C.c_void_p()
        check(load_library().wm_stream_create_cu_mask(words, n_words, C.byref(handle)), "wm_stream_create_cu_mask")
        _MASKED_STREAMS[key] = torch.cuda.ExternalStream(handle.value)
    return _MASKED_STREAMS[key]
