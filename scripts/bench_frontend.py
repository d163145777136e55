"""Micro-benchmark: device log-mel front end (wm_log_mel) for a batch of 30 s clips."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, whisper_utils as wu
for B in (1, 16, 256):
    audio = torch.randn(B, wu.N_SAMPLES, device="cuda") * 0.1
    for _ in range(2): wu.log_mel_spectrogram_device(audio)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): wu.log_mel_spectrogram_device(audio)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"B={B}: {ms:.3f} ms ({ms / B * 1e3:.1f} us per clip, {B * 0.97e9 / ms / 1e9:.1f} TFLOP/s fp32)", flush=True)
