"""WhisperEncoding: the encoder session wrapper.  Same class, constructor and methods as the
reference's examples/whisper/encoding.py (W/encoding.py:12-76), over the gfx950 engine.

`get_audio_features(mel)` takes fp16 `[B, n_mels, 2*n_audio_ctx]` on the GPU (the reference is
hard-wired to `[1, 80, 3000]`; any batch works here) and returns fp16
`[B, n_audio_ctx, n_audio_state]`.
"""
from __future__ import annotations

import itertools
import json
from collections import OrderedDict

import torch

import native
from build import get_engine_name
from session import Session, TensorInfo, str_dtype_to_trt, trt_dtype_to_torch, logger


# Every encoder run stamps its output tensor with a fresh generation number (`audio_features.wm_generation`).
# The engine writes through a raw pointer, which torch's `_version` counter never sees, so (pointer, shape,
# version) alone cannot tell two batches encoded into the same buffer apart; WhisperDecoding re-uses cross K/V
# only for a tensor that carries the generation it cached.
_GENERATION = itertools.count(1)

# CUs the encoder keeps when it runs beside a decode loop (WhisperEncoding.prefetch): measured at B = 576 on MI355X,
# see DESIGN.md "encoder under the decode loop"
DEFAULT_SHARED_CU_BUDGET = 96
PREFETCH_MIN_BATCH = 9           # prefetch() runs beside the decode loop from this many clips on; fewer: collect() runs the pass (see prefetch)
PREFETCH_QUEUE_DEFAULT = "pooled"      # hardware queue of the prefetched encoder pass: "pooled" | "dedicated" (WM_PREFETCH_QUEUE; _side_stream)


def stamp_generation(audio_features):
    audio_features.wm_generation = next(_GENERATION)
    return audio_features


class WhisperEncoding:
    def __init__(self, engine_dir, only_torch: bool = False):
        self.dtype = 'float16'
        if not only_torch:
            self.session = self.get_session(engine_dir)

    def get_session(self, engine_dir):
        config_path = engine_dir / 'encoder_config.json'
        with open(config_path, 'r') as f:
            config = json.load(f)
        self.use_gpt_attention_plugin = config['plugin_config']['gpt_attention_plugin']
        dtype = config['builder_config']['precision']
        world_size = config['builder_config']['tensor_parallel']
        self.num_heads = config['builder_config']['num_heads'] // world_size
        self.hidden_size = config['builder_config']['hidden_size'] // world_size
        self.num_layers = config['builder_config']['num_layers']
        self.dtype = dtype
        serialize_path = engine_dir / get_engine_name('whisper_encoder', self.dtype, world_size, 0)
        with open(serialize_path, 'rb') as f:
            session = Session.from_serialized_engine(f.read())
        return session

    def torch_get_audio_features(self, model, mel):
        """The PyTorch path (W/encoding.py:43-46): `model` is any module with `.encoder(mel)`."""
        with torch.no_grad():
            audio_features = model.encoder(mel)
        return audio_features

    def get_audio_features(self, mel):
        inputs = OrderedDict()
        output_list = []
        mel = mel.type(torch.float16).contiguous()
        inputs.update({'x': mel})
        output_list.append(TensorInfo('x', str_dtype_to_trt("float16"), mel.shape))
        # the two length tensors are dummies in the reference too (W/encoding.py:55-61)
        input_lengths = torch.ones((1,), dtype=torch.int32, device=mel.device)
        inputs.update({'input_lengths': input_lengths})
        output_list.append(TensorInfo('input_lengths', str_dtype_to_trt("int32"), input_lengths.shape))
        inputs.update({'max_input_length': input_lengths})
        output_list.append(TensorInfo('max_input_length', str_dtype_to_trt("int32"), input_lengths.shape))

        output_info = self.session.infer_shapes(output_list)
        logger.debug(f'output info {output_info}')
        outputs = {t.name: torch.empty(tuple(t.shape), dtype=trt_dtype_to_torch(t.dtype), device=mel.device)
                   for t in output_info}
        stream = torch.cuda.current_stream()
        ok = self.session.run(inputs=inputs, outputs=outputs, stream=stream.cuda_stream)
        assert ok, 'Engine execution failed'
        stream.synchronize()
        return stamp_generation(outputs['output'])

    def get_audio_features_async(self, mel, out=None, cu_budget: int = 0):
        """Fast path: no dictionaries, no synchronisation; enqueued on the current stream."""
        mel = mel.type(torch.float16).contiguous()
        d = self.session.dims
        if out is None:
            out = torch.empty((mel.shape[0], d['n_audio_ctx'], d['n_audio_state']), dtype=torch.float16, device=mel.device)
        self.session.encoder_forward(mel, out, torch.cuda.current_stream().cuda_stream, cu_budget)
        return stamp_generation(out)

    # -- the encoder of the NEXT batch under the decode loop of the current one --------------------------------------
    # The decode loop is HBM-bound (it streams the cross-attention K/V once per token), the encoder MFMA-bound.  With
    # cu_budget workgroups pinned to as many CUs the encoder leaves the rest of the chip to the decode kernels, which are
    # then never dispatched behind a 128 KB-LDS GEMM tile.  prefetch() enqueues from a helper thread on a stream of its
    # own (the ctypes calls release the GIL); collect() joins it and makes the current stream wait for the result.
    #
    # Round 5: the budget is given back when the loop ends.  A pass confined to 96 CUs takes three times the whole chip's time; beside
    # a decode loop of 128 tokens x 576 utterances that is the point, behind a loop that ended after 30 tokens (LibriSpeech-like
    # lengths) it made the pipelined schedule 34 % SLOWER than one stage after the other.  The helper thread therefore issues the
    # pass LAYER BY LAYER (wm_encoder_forward_range), each layer when the one before it has finished, and looks before each layer whether
    # the loop has ended -- `loop_ended()` records an event on the caller's stream behind the loop; the helper polls it -- from then
    # on the layers still to come are issued for the whole chip.  No prediction of the loop's length is needed, a short loop costs
    # at most the layer that is running, and the result is bit-identical whatever the cut (same tiles, other workgroups).
    def prefetch(self, mel, cu_budget: int = DEFAULT_SHARED_CU_BUDGET):
        import threading
        assert getattr(self, "_prefetch", None) is None, "one prefetch at a time"
        if int(mel.shape[0]) < getattr(self, "prefetch_min_batch", PREFETCH_MIN_BATCH):      # (an attribute: tests of the helper path set it to 1)
            # Round 6: batches of up to eight clips are NOT run beside the decode loop -- collect() runs the pass, on the caller's stream.
            # (1) It does not pay there: the loop of such a batch is the one-launch step, which wants every CU for itself (one stage after
            # the other is 0.5 % faster at one clip and 4 % at eight: profiles/r6a_*, r6z_*).  (2) It is not safe there: a one-launch step
            # dispatched within ~ 2 ms of the START of a budget-confined pass gave up its bounded waits in 5 of 25 runs of bench.py --batch 5
            # (profiles/r6o_*, r6p_*: the step's 256 workgroups must be resident together, the pass's first kernels take CUs away under it;
            # recovered by decoding again, but a second lost and the device off the form).  Rounds 4-5 never saw it because the eagerly
            # issued prefill kept the first step 2.7 ms behind the start of the pass.
            self._prefetch = (None, {"deferred": mel})
            return
        if getattr(self, "_prefetch_stream", None) is None:
            self._prefetch_stream = self._side_stream(mel.device)
        side, box = self._prefetch_stream, {"loop_done": None, "released_at": None}
        side.wait_stream(torch.cuda.current_stream())          # mel may still be in flight on the caller's stream
        device = mel.device

        timed = bool(getattr(self, "time_prefetch", False))        # bench.py: events around the prefetched pass and around collect()'s wait
        n_layer = self.session.dims['n_audio_layer']
        release = bool(getattr(self, "release_budget_when_loop_ends", True)) and cu_budget > 0

        def work():
            try:
                with native.CAPTURE_LOCK:
                    torch.cuda.set_device(device)
                with torch.cuda.stream(side):
                    with native.CAPTURE_LOCK:
                        if timed:
                            box["t0"] = torch.cuda.Event(enable_timing=True); box["t0"].record()
                        m16 = mel.type(torch.float16).contiguous()
                        d = self.session.dims
                        out = torch.empty((m16.shape[0], d['n_audio_ctx'], d['n_audio_state']), dtype=torch.float16, device=m16.device)
                    if not release:
                        with native.CAPTURE_LOCK:
                            self.session.encoder_forward(m16, out, side.cuda_stream, cu_budget)
                    else:
                        # layers per issue: a layer of B clips takes ~ 0.23 ms x B on a 96-CU budget and the helper thread needs ~ 0.1 ms to
                        # wake up and issue one -- small batches go in chunks of several layers (B = 1: two chunks; from 20 clips on: one layer)
                        budget, done, chunk = cu_budget, [], max(1, -(-20 // int(m16.shape[0])))
                        import time
                        for i in range(0, n_layer, chunk):
                            # a layer is issued when the one before it has finished: the look at the loop's event below is fresh (the GPU
                            # idles for the ~0.1-0.3 ms the host needs to notice and to issue a layer's 7 launches, 0.1-0.3 % of a 45-135 ms
                            # layer; two layers queued ahead cost a short loop 180 ms of budget-confined work behind its end: profiles/r5h_*).
                            # The wait POLLS, and every call into the runtime is made under native.CAPTURE_LOCK: the main thread may be
                            # capturing a decode step's graph, and a runtime call from this thread inside that window can invalidate it
                            while done:
                                with native.CAPTURE_LOCK:
                                    if done[-1].query():
                                        break
                                time.sleep(0.0002)
                            with native.CAPTURE_LOCK:
                                ev = box["loop_done"]
                                if budget > 0 and ev is not None and ev.query():
                                    budget, box["released_at"] = 0, i  # the loop has ended: the whole chip for what is left
                                self.session.encoder_forward_range(m16, out, side.cuda_stream, budget, i, min(n_layer, i + chunk))
                                e = torch.cuda.Event(); e.record()
                            done.append(e)
                    with native.CAPTURE_LOCK:
                        box["xa"] = stamp_generation(out)
                        if timed:
                            box["t1"] = torch.cuda.Event(enable_timing=True); box["t1"].record()
            except BaseException as exc:                       # re-raised by collect()
                box["error"] = exc

        th = threading.Thread(target=work, name="wm-encoder-prefetch", daemon=True)
        th.start()
        self._prefetch = (th, box)

    @staticmethod
    def _side_stream(device):
        """The stream the prefetched pass is issued on.  A torch pool stream shares one of ROCm's GPU_MAX_HW_QUEUES (4) hardware queues with
        whatever else the process created -- including the caller's current stream: the marker a decode loop's group streams wait for then sits
        BEHIND the chunk of encoder layers just issued.  Round 6 measured it (bench.py `pipeline.other_ms`, the span between two events on the
        caller's idle stream around prefetch(): 5.7 ms at one clip, where a chunk is 20 layers; 2.9 at two, 1.5 at eight; the first sampled
        token of a one-clip batch came 10.7 ms after its encoder output).  WM_PREFETCH_QUEUE=dedicated gives the pass a hardware queue of its
        own (a stream created with a full CU mask is never multiplexed: decoding.py `_group_streams`); `pooled` is the torch pool stream."""
        import os
        if os.environ.get("WM_PREFETCH_QUEUE", PREFETCH_QUEUE_DEFAULT) == "dedicated":
            try:
                n_cu = torch.cuda.get_device_properties(device).multi_processor_count
                with torch.cuda.device(device):
                    return native.create_masked_stream([True] * n_cu, 7)      # (index 7: the utterance groups use 0 .. 2)
            except native.WmError:
                pass
        return torch.cuda.Stream(device=device)

    def loop_ended(self):
        """Tell a pass in flight (prefetch) that the decode loop it runs beside has been issued to its end: an event behind the loop on the
        CURRENT stream (WhisperDecoding.main_loop joins its group streams into it before it returns); once the GPU has passed it, the
        encoder layers still to come take the whole chip.  Harmless without a pass in flight."""
        pending = getattr(self, "_prefetch", None)
        if pending is not None and pending[1].get("loop_done") is None:
            ev = torch.cuda.Event()
            ev.record()
            pending[1]["loop_done"] = ev

    def collect(self):
        """The audio features of the batch handed to prefetch(); the current stream waits for them."""
        self.loop_ended()                        # (a caller that did not say so: whatever it ran beside the pass lies before this point)
        th, box = self._prefetch
        self._prefetch = None
        if th is None:                           # a small batch (prefetch): the pass runs now, on the caller's stream, with the whole chip
            timed = bool(getattr(self, "time_prefetch", False))
            if timed:
                w0 = torch.cuda.Event(enable_timing=True); w0.record()
            xa = self.get_audio_features_async(box["deferred"])
            if timed:
                w1 = torch.cuda.Event(enable_timing=True); w1.record()
                self.prefetch_events = getattr(self, "prefetch_events", [])
                self.prefetch_events.append((w0, w1, w0, w1))
            self.last_release_layer = 0
            return xa
        th.join()
        if "error" in box:
            raise box["error"]
        cur = torch.cuda.current_stream()
        if "t0" in box:                          # (encoder start, encoder end, before the wait, after the wait): read after a synchronize
            w0 = torch.cuda.Event(enable_timing=True); w0.record()
        cur.wait_stream(self._prefetch_stream)
        if "t0" in box:
            w1 = torch.cuda.Event(enable_timing=True); w1.record()
            self.prefetch_events = getattr(self, "prefetch_events", [])
            self.prefetch_events.append((box["t0"], box["t1"], w0, w1))
        self.last_release_layer = box.get("released_at")        # layer from which the pass had the whole chip (None: budget to the end)
        box["xa"].record_stream(cur)             # allocated on the side stream, used on this one
        return box["xa"]
