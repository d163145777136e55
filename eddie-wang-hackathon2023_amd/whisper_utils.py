"""Audio front-end: PCM loading, pad/trim, log-mel spectrogram.  Mirror of W/whisper_utils.py:17-146
(`load_audio`, `pad_or_trim`, `mel_filters`, `log_mel_spectrogram`, same constants).

Differences: the reference shells out to ffmpeg (absent here) -- `load_audio` reads 16 kHz mono PCM16
`.wav` with the standard library and `.npy` waveforms; the 80x201 mel filterbank is regenerated with
the recipe the reference quotes (`librosa.filters.mel(sr=16000, n_fft=400, n_mels=80)`,
W/whisper_utils.py:85-90: Slaney scale, Slaney area normalisation) instead of shipping the `.npz`;
tests/golden/mel.npz (made with the reference's own function and asset) pins both.  The STFT runs
through torch on whatever device the audio is on; a fused HIP STFT+mel kernel is SURVEY 8f-1 (next).
"""
from __future__ import annotations

import wave
from functools import lru_cache
from typing import Optional, Union

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 400
N_MELS = 80
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE  # 480000 samples in a 30-second chunk
N_FRAMES = N_SAMPLES // HOP_LENGTH       # 3000 frames in a mel spectrogram input


def decode_flac(data: bytes, verify_md5: bool = True):
    """FLAC bytes -> (int32 samples [n, channels], sample_rate, bits_per_sample) through the native decoder
    (wm_flac_decode, csrc/flac_decode.hip; frame CRCs checked there).  With `verify_md5` the decoded PCM is
    hashed and compared with the stream's own STREAMINFO signature (skipped when the encoder left it zero)."""
    import ctypes as C
    import hashlib
    import native
    lib = native.load_library()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    info = native.WmFlacStreamInfo()
    native.check(lib.wm_flac_info(buf, len(data), C.byref(info)), "wm_flac_info")
    capacity = int(info.total_samples)
    if capacity == 0:      # unknown length: every frame holds at most max_block_size samples and at least 11 bytes
        capacity = (len(data) // 11 + 1) * max(int(info.max_block_size), 16)
    pcm = np.empty((capacity, info.channels), dtype=np.int32)
    n = C.c_int64(0)
    native.check(lib.wm_flac_decode(buf, len(data), pcm.ctypes.data, capacity, C.byref(n)), "wm_flac_decode")
    pcm = pcm[: n.value]
    signature = bytes(info.md5)
    if verify_md5 and any(signature):
        width = (info.bits_per_sample + 7) // 8
        raw = pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :width].tobytes()
        if hashlib.md5(raw).digest() != signature:
            raise RuntimeError("FLAC MD5 signature mismatch: decoded audio differs from what the encoder saw")
    return pcm, int(info.sample_rate), int(info.bits_per_sample)


def load_audio(file: str, sr: int = SAMPLE_RATE) -> np.ndarray:
    """Mono float32 waveform in [-1, 1] at `sr` Hz (W/whisper_utils.py:17-54, which pipes every file
    through ffmpeg).  Without ffmpeg: PCM16 .wav, .flac (own decoder) and float .npy, already at `sr` Hz;
    several channels are averaged."""
    file = str(file)
    if file.endswith(".npy"):
        return np.load(file).astype(np.float32).flatten()
    if file.endswith(".flac"):
        with open(file, "rb") as f:
            pcm, rate, bits = decode_flac(f.read())
        if rate != sr:
            raise RuntimeError(f"{file}: {rate} Hz, need {sr} Hz (no resampler without ffmpeg)")
        return (pcm.astype(np.float32).mean(axis=1) / float(1 << (bits - 1))).astype(np.float32)
    if not file.endswith(".wav"):
        raise RuntimeError(f"cannot decode {file}: only .wav (PCM16) / .flac / .npy are supported without ffmpeg")
    with wave.open(file, "rb") as w:
        if w.getsampwidth() != 2 or w.getframerate() != sr:
            raise RuntimeError(f"{file}: need 16-bit PCM at {sr} Hz, got {8 * w.getsampwidth()}-bit at {w.getframerate()} Hz")
        pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16).reshape(-1, w.getnchannels())
    return pcm.astype(np.float32).mean(axis=1) / 32768.0


def pad_or_trim(array, length: int = N_SAMPLES, *, axis: int = -1):
    """Pad with zeros or cut to `length` samples along `axis` (W/whisper_utils.py:56-79)."""
    if torch.is_tensor(array):
        n = array.shape[axis]
        if n > length:
            array = array.narrow(axis, 0, length)
        elif n < length:
            pad = [0, 0] * array.ndim
            pad[2 * (array.ndim - 1 - (axis % array.ndim)) + 1] = length - n
            array = F.pad(array, pad)
        return array
    n = array.shape[axis]
    if n > length:
        array = np.take(array, np.arange(length), axis=axis)
    elif n < length:
        widths = [(0, 0)] * array.ndim
        widths[axis] = (0, length - n)
        array = np.pad(array, widths)
    return array


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


@lru_cache(maxsize=None)
def _mel_filterbank(n_mels: int = N_MELS, n_fft: int = N_FFT, sr: int = SAMPLE_RATE) -> np.ndarray:
    fft_freqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def mel_filters(device, n_mels: int = N_MELS) -> torch.Tensor:
    assert n_mels == 80, f"Unsupported n_mels: {n_mels}"
    return torch.from_numpy(_mel_filterbank(n_mels)).to(device)


def log_mel_spectrogram(audio: Union[str, np.ndarray, torch.Tensor], n_mels: int = N_MELS, padding: int = 0,
                        device: Optional[Union[str, torch.device]] = None) -> torch.Tensor:
    """[80, n_frames] log-mel: Hann STFT (n_fft 400, hop 160), |.|^2, mel projection, log10, clamp to
    max - 8, (x + 4) / 4  (W/whisper_utils.py:99-146)."""
    if not torch.is_tensor(audio):
        if isinstance(audio, str):
            audio = load_audio(audio)
        audio = torch.from_numpy(audio)
    if device is not None:
        audio = audio.to(device)
    if padding > 0:
        audio = F.pad(audio, (0, padding))
    window = torch.hann_window(N_FFT).to(audio.device)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    mel_spec = mel_filters(audio.device, n_mels) @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def log_mel_spectrogram_device(audio: torch.Tensor, n_mels: int = N_MELS, dtype: torch.dtype = torch.float16,
                               stream: Optional[int] = None) -> torch.Tensor:
    """The same transform as a HIP kernel (`wm_log_mel`, csrc/frontend.hip): `audio` fp32 [B, n] or [n]
    on the GPU, already padded / trimmed (n a multiple of HOP_LENGTH) -> [B, n_mels, n // HOP_LENGTH]
    (or [n_mels, frames] for a 1-D input) in `dtype` (fp16: what the encoder engine takes; fp32: what
    log_mel_spectrogram returns).  Each clip is clamped to its own max - 8, i.e. the reference applied
    per clip.  No CPU fallback: without the native library this raises."""
    import ctypes as C
    import native
    assert audio.is_cuda and audio.dtype == torch.float32, "audio must be an fp32 tensor on the GPU"
    assert dtype in (torch.float16, torch.float32)
    single = audio.dim() == 1
    a = (audio[None] if single else audio).contiguous()
    lib = native.load_library()
    B, n = a.shape
    filt = mel_filters(a.device, n_mels).float().contiguous()
    out = torch.empty((B, n_mels, n // HOP_LENGTH), dtype=dtype, device=a.device)
    ws_bytes = lib.wm_log_mel_workspace_bytes(B, n, n_mels)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
    s = torch.cuda.current_stream(a.device).cuda_stream if stream is None else stream
    o16 = out.data_ptr() if dtype == torch.float16 else None
    o32 = out.data_ptr() if dtype == torch.float32 else None
    native.check(lib.wm_log_mel(a.data_ptr(), B, n, a.stride(0), filt.data_ptr(), n_mels, o16, o32, ws.data_ptr(),
                                C.c_size_t(ws_bytes), s), "wm_log_mel")
    return out[0] if single else out
