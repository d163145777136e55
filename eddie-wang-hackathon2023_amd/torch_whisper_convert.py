"""torch_whisper_convert.py: int8 KV-cache calibration -> `quantize/1-gpu/*.bin`.

Same CLI and the same output files as the reference tool (W/torch_whisper_convert.py:50-110,121-216;
W = /root/reference/tensorrt_llm_july-release-v1/examples/whisper):

    python torch_whisper_convert.py -i large-v2.pt -o quantize -kv [--dataset_dir D] [--engine_dir E]

writes `<out>/1-gpu/model.decoder.blocks.{i}.attn.query_key_value.scale_y_quant_orig.bin`
(fp32[1]) and `config.ini`, the files `build.py --int8_kv_cache` reads (W/weight.py:236-243).

Statistic: for every decoder layer t = max(|Q_out|, |K_out|, |V_out|) / 127 over the
outputs of the three self-attention projections while the fp16 model greedily decodes the
calibration clips (language-ID pass + main loop, like W/smoothquant.py:147-170).  The reference
collects it with forward hooks on its PyTorch model; here the fp16 ENGINE decodes and the
self-attention kernel itself keeps the running maximum (wm_decoder_io.qkv_amax, csrc/attn_decode.hip),
so calibration runs at engine speed.  Q outputs are included on purpose -- that is what the
reference's merged q/k/v range does (W/torch_whisper_convert.py:145-167).

Calibration input: `--dataset_dir` as the reference uses it (W/torch_whisper_convert.py:121-216 over
`LibriSpeech/valid-clean`): every `.flac` below the directory, decoded by wm_flac_decode, padded / trimmed to
30 s and turned into log-mels by wm_log_mel on the GPU (the reference shells out to ffmpeg, W/whisper_utils.py:17-54);
`.npy` log-mels `[80, 3000]` are accepted too; or `--synthetic_clips N` seeded synthetic mels.
"""
from __future__ import annotations

import argparse
import configparser
import os
import tempfile
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch


def scale_file_name(layer: int) -> str:
    return f"model.decoder.blocks.{layer}.attn.query_key_value.scale_y_quant_orig.bin"


def capture_kv_activation_range(engine_dir, mels: torch.Tensor, batch: int = 8, sample_len: Optional[int] = None,
                                ignore_eot: bool = False, token_log: Optional[list] = None) -> List[float]:
    """max(|q|,|k|,|v|) per decoder layer while the engines in `engine_dir` (fp16 KV) decode `mels`
    [N, n_mels, 2*n_audio_ctx].  Mirror of capture_activation_range (W/smoothquant.py:117-175)
    restricted to what `-kv` consumes.  `token_log` (tests): receives, per batch, the token rows the loop
    decoded (int64 [b, n]) and the length of the start sequence, i.e. the exact token path the statistic saw."""
    from decoding import WhisperDecoding
    from encoding import WhisperEncoding
    engine_dir = Path(engine_dir)
    enc, dec = WhisperEncoding(engine_dir), WhisperDecoding(engine_dir)
    assert not dec.use_int8_kv_cache, "calibrate with an fp16-KV engine"
    if sample_len is not None:
        dec.sample_len = sample_len
    n_layer = dec.decoder_config['num_layers']
    amax = torch.zeros(n_layer, dtype=torch.float32, device='cuda')
    dec.decoder_session.qkv_amax = amax
    try:
        for i in range(0, mels.shape[0], batch):
            mel = mels[i:i + batch].to('cuda').type(torch.float16)
            xa = enc.get_audio_features(mel)
            dec.detect_language(xa)
            tokens, _, _ = dec.main_loop(xa, ignore_eot=ignore_eot)
            if token_log is not None:
                token_log.append((tokens.cpu(), dec.initial_token_length, dec.tokenizer.sot if dec.is_multilingual else None))
        torch.cuda.synchronize()
    finally:
        dec.decoder_session.qkv_amax = None
    return amax.cpu().tolist()


def capture_cross_kv_range(engine_dir, mels: torch.Tensor, batch: int = 8) -> List[float]:
    """max(|K|, |V|) of every layer's cross-attention K/V over `mels`, from the fp16 engines in `engine_dir`: the
    statistic the opt-in int8 cross-K/V mode (build.py --int8_cross_kv, beyond the reference) is scaled by."""
    from decoding import WhisperDecoding
    from encoding import WhisperEncoding
    engine_dir = Path(engine_dir)
    enc, dec = WhisperEncoding(engine_dir), WhisperDecoding(engine_dir)
    assert not dec.use_int8_cross_kv, "calibrate with an fp16 cross-K/V engine"
    amax = [0.0] * dec.decoder_config['num_layers']
    for i in range(0, mels.shape[0], batch):
        xa = enc.get_audio_features(mels[i:i + batch].to('cuda').type(torch.float16))
        for l, c in enumerate(dec.xa2cross_key_value(xa)):
            amax[l] = max(amax[l], float(c.abs().max()))
    return amax


def cross_scale_file_name(layer: int) -> str:
    return f"model.decoder.blocks.{layer}.cross_attn.key_value.scale_y_quant_orig.bin"


def write_cross_kv_scales(out_dir, amax: List[float]) -> Path:
    saved_dir = Path(out_dir) / "1-gpu"
    saved_dir.mkdir(parents=True, exist_ok=True)
    for i, a in enumerate(amax):
        np.array([np.float32(a) / np.float32(127.0)], dtype=np.float32).tofile(saved_dir / cross_scale_file_name(i))
    return saved_dir


def write_kv_scales(out_dir, amax: List[float], meta: Optional[dict] = None) -> Path:
    """scale_y_quant_orig = max / 127 (W/utils/convert.py:76-78,98,138-140), one fp32 per layer."""
    saved_dir = Path(out_dir) / "1-gpu"
    saved_dir.mkdir(parents=True, exist_ok=True)
    for i, a in enumerate(amax):
        np.array([np.float32(a) / np.float32(127.0)], dtype=np.float32).tofile(saved_dir / scale_file_name(i))
    config = configparser.ConfigParser()
    config["whisper"] = {k: str(v) for k, v in (meta or {}).items()}
    with open(saved_dir / "config.ini", "w") as f:
        config.write(f)
    return saved_dir


def parse_arguments(args=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument('--out-dir', '-o', type=str, help='file name of output directory', required=True)
    parser.add_argument('--in-file', '-i', type=str, help='checkpoint (.pt); omit with --synthetic', default=None)
    parser.add_argument('--tensor-parallelism', '-tp', type=int, default=1)
    parser.add_argument('--processes', '-p', type=int, default=1)
    parser.add_argument('--calibrate-kv-cache', '-kv', default=False, action="store_true",
                        help='Generate scaling factors for KV cache. Used for storing KV cache in int8.')
    parser.add_argument('--smoothquant', '-sq', type=float, default=None,
                        help='accepted for CLI parity; SmoothQuant is GPT-only in the reference (quant.py:8-10)')
    parser.add_argument('--model', default="whisper", type=str)
    parser.add_argument('--storage-type', '-t', type=str, default="float16", choices=["float32", "float16", "bfloat16"])
    parser.add_argument('--dataset_dir', type=str, default='./LibriSpeech/valid-clean')
    parser.add_argument('--engine_dir', type=str, default=None,
                        help='fp16-KV engine directory to calibrate with (built on the fly when omitted)')
    parser.add_argument('--synthetic', type=str, default=None, help='random-init checkpoint size instead of --in-file')
    parser.add_argument('--synthetic_clips', type=int, default=0, help='calibrate on N seeded synthetic mels')
    parser.add_argument('--seed', type=int, default=0)
    return parser.parse_args(args)


def load_calibration_mels(args, dims) -> torch.Tensor:
    import synthetic
    if args.synthetic_clips > 0:
        return synthetic.synthetic_mel(args.synthetic_clips, 2 * dims['n_audio_ctx'], dims['n_mels'], 4321)
    root = Path(args.dataset_dir)
    flacs = sorted(root.rglob('*.flac'))
    if flacs:
        import whisper_utils as wu
        mels = []
        for f in flacs:
            audio = torch.from_numpy(wu.pad_or_trim(wu.load_audio(str(f)))).cuda()
            mels.append(wu.log_mel_spectrogram_device(audio, n_mels=dims['n_mels'], dtype=torch.float16).cpu())
        return torch.stack(mels)
    files = sorted(p for p in root.rglob('*.npy'))
    if not files:
        raise FileNotFoundError(f"no .flac audio or .npy log-mels below {args.dataset_dir}")
    return torch.stack([torch.from_numpy(np.load(f)).half() for f in files])


def run_conversion(args):
    import build as B
    import synthetic
    if not args.calibrate_kv_cache:
        raise SystemExit("nothing to do: only -kv (int8 KV-cache calibration) applies to Whisper")
    model = synthetic.synthetic_checkpoint(args.synthetic, args.seed) if args.synthetic else \
        torch.load(args.in_file, map_location='cpu')
    with tempfile.TemporaryDirectory() as tmp:
        engine_dir = args.engine_dir
        if engine_dir is None:      # the reference calibrates the un-quantised fp16 model
            engine_dir = os.path.join(tmp, "calib_engine")
            B.build_from_checkpoint(model, B.parse_arguments(["--output_dir", engine_dir, "--log_level", "error"]))
        mels = load_calibration_mels(args, model['dims'])
        amax = capture_kv_activation_range(engine_dir, mels)
        cross_amax = capture_cross_kv_range(engine_dir, mels)
    out = write_kv_scales(args.out_dir, amax, {k: v for k, v in vars(args).items()})
    write_cross_kv_scales(args.out_dir, cross_amax)          # for the opt-in int8 cross-K/V mode (build.py --int8_cross_kv)
    print(f"wrote {len(amax)} KV scales (+ cross-K/V scales) to {out}")


if __name__ == "__main__":
    run_conversion(parse_arguments())
