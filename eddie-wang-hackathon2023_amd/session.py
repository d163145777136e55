"""`Session` / `TensorInfo`: the object the Whisper wrappers hold, re-implemented over the C ABI.

Same surface as R/tensorrt_llm/runtime/session.py:27-31,53-61,116-178 (R = /root/reference/
tensorrt_llm_july-release-v1): `Session.from_serialized_engine(bytes)`, `infer_shapes(List[
TensorInfo]) -> List[TensorInfo] | None` (logs and returns None on an unknown name or a wrong
dtype), `run(inputs, outputs, stream) -> bool` -- asynchronous enqueue on the caller's stream, the
caller owns and pre-allocates every input and output tensor, and synchronises.

Engine I/O by tensor name is the reference's (whisper/model.py:171-197,
301-467,543-555), generalised from batch 1 / large-v2 to any batch and model size.  dtypes are
spelled as strings ("float16", "int32", "int8", "float32"): `str_dtype_to_trt` is the identity
here, there is no TensorRT enum to map to.

Besides the by-name `run` there are typed fast-path methods (`encoder_forward`, `cross_kv`,
`decoder_step`) that skip the dictionaries and allow in-place KV-cache append.
"""
from __future__ import annotations

import ctypes as C
import itertools
import logging
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import torch

import native
from native import Engine, WmDecoderIO, check, ptr_array

logger = logging.getLogger("whisper_mi355")

ENGINE_ENCODER, ENGINE_DECODER, ENGINE_CROSS_KV = 0, 1, 2
FLAG_WEIGHT_ONLY_INT8, FLAG_INT8_KV, FLAG_GELU_TANH, FLAG_INT8_CROSS_KV = 1, 2, 4, 16

_STR_TO_TORCH = {"float16": torch.float16, "float32": torch.float32, "int32": torch.int32, "int8": torch.int8,
                 "bfloat16": torch.bfloat16}


def str_dtype_to_trt(dtype: str) -> str:
    """Identity: engine dtypes are plain strings (reference: tensorrt_llm._utils.str_dtype_to_trt)."""
    if dtype not in _STR_TO_TORCH:
        raise ValueError(f"unsupported dtype {dtype}")
    return dtype


def str_dtype_to_torch(dtype: str) -> torch.dtype:
    return _STR_TO_TORCH[dtype]


def trt_dtype_to_torch(dtype: str) -> torch.dtype:
    return _STR_TO_TORCH[dtype]


_WORKSPACE_IDS = itertools.count(1)      # identities of workspace allocations (wm_decoder_io.workspace_id), process-wide


@dataclass
class TensorInfo:
    name: str
    dtype: str
    shape: tuple


class Session(object):
    def __init__(self, **kwargs):
        # use Session.from_serialized_engine to create a session
        pass

    def _init(self, engine_buffer, device: Optional[int] = None):
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self._engine = Engine(engine_buffer, device)
        self._device = device
        self._workspaces: Dict[tuple, torch.Tensor] = {}
        self._shapes: Dict[str, tuple] = {}
        self.qkv_amax: Optional[torch.Tensor] = None
        return self

    @staticmethod
    def from_serialized_engine(engine, device: Optional[int] = None) -> "Session":
        """Create a session from the bytes of one *.engine file (session.py:53-61)."""
        return Session()._init(engine, device)

    # -- properties ----------------------------------------------------------------------------
    @property
    def engine(self) -> Engine:
        return self._engine

    @property
    def kind(self) -> int:
        return self._engine.kind

    @property
    def dims(self) -> dict:
        return self._engine.dims

    @property
    def kv_dtype(self) -> str:
        return "int8" if self._engine.flags & FLAG_INT8_KV else "float16"

    @property
    def cross_kv_dtype(self) -> str:
        """fp16 like the reference, or int8 codes for engines built with --int8_cross_kv (opt-in, beyond the reference)."""
        return "int8" if self._engine.flags & FLAG_INT8_CROSS_KV else "float16"

    def _workspace(self, key: tuple, nbytes: int) -> torch.Tensor:
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            # as the allocator hands it back, like the buffers the reference's session gets (torch.empty): the decoder's one-launch step
            # keeps a call counter and tagged granules in its workspace (csrc/gemv_chain.hip), and the LIBRARY initialises them, on the
            # stream of the first call that carries this allocation's id (a zero-fill here ran on torch's current stream, un-ordered
            # with the side stream the first call ran on)
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=f"cuda:{self._device}")
            ws._wm_id = next(_WORKSPACE_IDS)      # wm_decoder_io.workspace_id: a new identity for every allocation
            self._workspaces[key] = ws
        return ws

    # -- by-name interface -----------------------------------------------------------------------
    def _input_spec(self) -> Dict[str, Optional[str]]:
        """name -> dtype of every input this engine accepts (None = any: ignored dummies)."""
        d = self.dims
        if self.kind == ENGINE_ENCODER:
            return {"x": "float16", "input_lengths": "int32", "max_input_length": "int32"}
        if self.kind == ENGINE_CROSS_KV:
            return {"xa": "float16"}
        spec = {"x": "int32", "input_lengths": "int32", "max_input_length": "int32",
                "positional_embedding": "float16", "mask": "float32", "masked_tokens": "int32",
                "cache_indirection": "int32", "past_key_value_length": "int32", "sequence_length": "int32"}
        for i in range(d["n_text_layer"]):
            spec[f"past_key_value_{i}"] = self.kv_dtype
            spec[f"cross_past_key_value_{i}"] = self.cross_kv_dtype
        return spec

    def infer_shapes(self, inputs: List[TensorInfo], context=None) -> Optional[List[TensorInfo]]:
        """Record the input shapes and return the output TensorInfos (session.py:116-146)."""
        spec = self._input_spec()
        for i in inputs:
            if i.name not in spec:
                logger.error(f"Tensor:{i.name} is not an input tensor")
                return None
            if spec[i.name] is not None and spec[i.name] != i.dtype:
                logger.error(f"Tensor:{i.name} has wrong dtype")
                return None
            self._shapes[i.name] = tuple(int(s) for s in i.shape)
        d = self.dims
        if self.kind == ENGINE_ENCODER:
            b = self._shapes["x"][0]
            return [TensorInfo("output", "float16", (b, d["n_audio_ctx"], d["n_audio_state"]))]
        if self.kind == ENGINE_CROSS_KV:
            b = self._shapes["xa"][0]
            shape = (b, 2, d["n_text_head"], d["n_audio_ctx"], d["n_text_state"] // d["n_text_head"])
            return [TensorInfo(f"cross_present_key_value_{i}", self.cross_kv_dtype, shape) for i in range(d["n_text_layer"])]
        b, l = self._shapes["x"]
        t = self._shapes.get("past_key_value_0", (b, 2, d["n_text_head"], 0, 64))[3]
        outs = [TensorInfo("output", "float16", (b, l, d["n_vocab"]))]
        outs += [TensorInfo(f"present_key_value_{i}", self.kv_dtype, (b, 2, d["n_text_head"], t + l, 64))
                 for i in range(d["n_text_layer"])]
        return outs

    def run(self, inputs: Dict[str, Any], outputs: Dict[str, Any], stream, context=None) -> bool:
        """Enqueue the engine on `stream` (session.py:148-178).  Returns False (and logs) on failure,
        like execute_async_v3 would."""
        try:
            if self.kind == ENGINE_ENCODER:
                mel = inputs["x"]
                self.encoder_forward(mel, outputs["output"], stream)
            elif self.kind == ENGINE_CROSS_KV:
                n = self.dims["n_text_layer"]
                self.cross_kv(inputs["xa"], [outputs[f"cross_present_key_value_{i}"] for i in range(n)], stream)
            else:
                n = self.dims["n_text_layer"]
                x = inputs["x"]
                b, l = x.shape
                t = self._shapes.get("past_key_value_0", (0, 0, 0, 0, 0))[3]
                past = [inputs[f"past_key_value_{i}"] for i in range(n)] if t > 0 else None
                self.decoder_step(x, inputs["positional_embedding"],
                                  [inputs[f"cross_past_key_value_{i}"] for i in range(n)],
                                  past, t, [outputs[f"present_key_value_{i}"] for i in range(n)], t + l,
                                  outputs["output"], t, stream)
            return True
        except (native.WmError, KeyError) as e:
            logger.error(f"Engine execution failed: {e}")
            return False

    # -- typed fast path ---------------------------------------------------------------------------
    def encoder_forward(self, mel: torch.Tensor, out: torch.Tensor, stream: int, cu_budget: int = 0):
        """cu_budget > 0: the pass runs on at most that many CUs, beside other streams' work (wm_encoder_forward_shared)."""
        lib = self._engine.lib
        b = mel.shape[0]
        nbytes = lib.wm_encoder_workspace_bytes(self._engine.handle, b)
        ws = self._workspace(("enc",), nbytes)        # one buffer, grown on demand: 26 GB at B = 576
        check(lib.wm_encoder_forward_shared(self._engine.handle, mel.data_ptr(), b, out.data_ptr(), ws.data_ptr(),
                                            ws.numel(), int(cu_budget), stream), "wm_encoder_forward")

    def encoder_forward_range(self, mel: torch.Tensor, out: torch.Tensor, stream: int, cu_budget: int, layer_begin: int, layer_end: int):
        """Layers [layer_begin, layer_end) of the pass (wm_encoder_forward_range): the convolutions with a range that starts at 0, the
        final LayerNorm with one that ends at n_audio_layer; the residual stream stays in the session's encoder workspace in between."""
        lib = self._engine.lib
        b = mel.shape[0]
        ws = self._workspace(("enc",), lib.wm_encoder_workspace_bytes(self._engine.handle, b))
        check(lib.wm_encoder_forward_range(self._engine.handle, mel.data_ptr(), b, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                           int(cu_budget), int(layer_begin), int(layer_end), stream), "wm_encoder_forward_range")

    def cross_kv(self, xa: torch.Tensor, outs: Sequence[torch.Tensor], stream: int):
        lib = self._engine.lib
        b = xa.shape[0]
        ws = self._workspace(("ckv",), lib.wm_cross_kv_workspace_bytes(self._engine.handle, b))
        check(lib.wm_cross_kv(self._engine.handle, xa.data_ptr(), b, ptr_array(outs), ws.data_ptr(), ws.numel(),
                              stream), "wm_cross_kv")

    def make_decoder_io(self, tokens: torch.Tensor, pos: torch.Tensor, cross: Sequence[torch.Tensor],
                        past: Optional[Sequence[torch.Tensor]], past_capacity: int,
                        present: Sequence[torch.Tensor], present_capacity: int, logits: torch.Tensor,
                        n_past: int, qkv_amax: Optional[torch.Tensor] = None, slot: int = 0,
                        n_past_dev: Optional[torch.Tensor] = None, n_new: Optional[int] = None,
                        live_rows: Optional[torch.Tensor] = None, not_alone: bool = False) -> WmDecoderIO:
        """The wm_decoder_io of one call.  tokens int32 [B, L] (any row stride: a column window of a wider
        buffer works); past/present per layer [B,2,H,capacity,64]; present may be the same tensors as past
        (in-place append).  The struct keeps its pointer arrays alive (`io._keep`)."""
        lib = self._engine.lib
        b, l = tokens.shape
        if n_new is not None:          # device step counter: `tokens` is the whole [B, capacity] buffer
            l = n_new
        assert tokens.dtype == torch.int32 and tokens.stride(1) == 1
        for name, group in (("cross", cross), ("present", present), ("past", past or ())):
            for t in group:
                if t.shape[0] != b:     # raw pointers cross the C ABI: a short buffer would be read out of bounds
                    raise native.WmError(f"decoder step: {name} K/V holds {t.shape[0]} utterances, tokens hold {b}")
        if logits.shape[0] != b:
            raise native.WmError(f"decoder step: logits hold {logits.shape[0]} utterances, tokens hold {b}")
        ws = self._workspace(("dec", b, l, slot), lib.wm_decoder_workspace_bytes(self._engine.handle, b, l))
        io = WmDecoderIO()
        io.batch, io.n_new, io.n_past = b, l, n_past
        io.tokens, io.positional_embedding = tokens.data_ptr(), pos.data_ptr()
        io.tokens_ld = tokens.stride(0)
        past_arr = ptr_array(past) if past is not None else None
        present_arr, cross_arr = ptr_array(present), ptr_array(cross)
        io.past = C.cast(past_arr, C.POINTER(C.c_void_p)) if past_arr is not None else None
        io.past_capacity = past_capacity
        io.present = C.cast(present_arr, C.POINTER(C.c_void_p))
        io.present_capacity = present_capacity
        io.cross = C.cast(cross_arr, C.POINTER(C.c_void_p))
        io.logits = logits.data_ptr()
        io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel()
        io.workspace_id = getattr(ws, "_wm_id", 0)
        if qkv_amax is None:
            qkv_amax = self.qkv_amax          # calibration hook set by torch_whisper_convert.py
        io.qkv_amax = qkv_amax.data_ptr() if qkv_amax is not None else None
        io.n_past_dev = n_past_dev.data_ptr() if n_past_dev is not None else None
        if live_rows is not None:      # int32 [1 + B]: count, then the rows still decoding (wm_step_finish keeps it current)
            assert live_rows.dtype == torch.int32 and live_rows.numel() >= 1 + b and live_rows.is_contiguous()
        io.live_rows = live_rows.data_ptr() if live_rows is not None else None
        io.not_alone = 1 if not_alone else 0      # other groups' steps in flight beside this one: never a one-launch form (whisper_mi355.h)
        io._keep = (past_arr, present_arr, cross_arr, ws, live_rows)
        return io

    def decoder_step(self, tokens, pos, cross, past, past_capacity, present, present_capacity, logits, n_past,
                     stream: int, qkv_amax=None, slot: int = 0, n_past_dev=None, n_new=None, live_rows=None, not_alone: bool = False):
        io = self.make_decoder_io(tokens, pos, cross, past, past_capacity, present, present_capacity, logits, n_past,
                                  qkv_amax, slot, n_past_dev, n_new, live_rows, not_alone)
        check(self._engine.lib.wm_decoder_step(self._engine.handle, C.byref(io), stream), "wm_decoder_step")

    def decoder_step_multi(self, ios: Sequence[WmDecoderIO], light_streams: Sequence[int], heavy_stream: int):
        """One decode step of several utterance groups, interleaved layer by layer (wm_decoder_step_multi):
        each group's short kernels on its own light stream, all cross-attention kernels on `heavy_stream`."""
        n = len(ios)
        io_ptrs = (C.POINTER(WmDecoderIO) * n)(*[C.pointer(io) for io in ios])
        streams = (C.c_void_p * n)(*light_streams)
        check(self._engine.lib.wm_decoder_step_multi(self._engine.handle, n, io_ptrs, streams, heavy_stream),
              "wm_decoder_step_multi")
