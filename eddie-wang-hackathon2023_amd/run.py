"""run.py: transcribe one input with a built engine directory.  Same CLI and output as the
reference (W/run.py:21-63): prints "transcribe time <s>" and the text.

Audio front-end: the reference shells out to ffmpeg and computes the log-mel on the GPU with
torch.stft (W/whisper_utils.py:17-146).  ffmpeg is not on these boxes; `--input_file` accepts a
`.flac` (decoded on the device: wm_flac_decode, csrc/flac_decode.hip -- LibriSpeech's format), a
16 kHz mono `.wav` (PCM16, standard library), a `.npy` log-mel `[80, 3000]` or the keyword
`synthetic`; the log-mel itself is the HIP front end (wm_log_mel, csrc/frontend.hip: STFT + mel +
log on the device), held to the reference's golden mel in
tests/test_gpu_model.py.  m4a / resampling (ffmpeg's job in the reference) stay out of scope.
"""
from __future__ import annotations

import argparse
import logging
import time
from pathlib import Path

import numpy as np
import torch

from decoding import WhisperDecoding
from encoding import WhisperEncoding


def parse_arguments():
    parser = argparse.ArgumentParser()
    parser.add_argument('--log_level', type=str, default='error')
    parser.add_argument('--engine_dir', type=str, default='whisper_outputs')
    parser.add_argument('--input_file', type=str, default='synthetic')
    parser.add_argument('--vocab', type=str, default=None, help='path to multilingual.tiktoken (text output)')
    return parser.parse_args()


def load_mel(input_file: str) -> torch.Tensor:
    if input_file == 'synthetic':
        import synthetic
        return synthetic.synthetic_mel(1)[0].float()
    if input_file.endswith('.npy'):
        return torch.from_numpy(np.load(input_file)).float()
    import whisper_utils
    audio = whisper_utils.pad_or_trim(whisper_utils.load_audio(input_file))
    # STFT + mel projection on the GPU (wm_log_mel); the torch.stft path stays in whisper_utils as the CPU mirror
    return whisper_utils.log_mel_spectrogram_device(torch.from_numpy(audio).float().cuda(), dtype=torch.float32)


def generate(log_level: str = 'error', engine_dir: str = 'whisper_outputs', input_file: str = 'synthetic',
             vocab: str = None):
    logging.basicConfig(level=getattr(logging, log_level.upper(), logging.ERROR))
    torch.cuda.set_device(0)
    mel = load_mel(input_file).to('cuda').type(torch.float16).unsqueeze(0)
    engine_dir = Path(engine_dir)
    whisper_encoding = WhisperEncoding(engine_dir)
    whisper_decoding = WhisperDecoding(engine_dir, vocab_path=vocab)
    begin_time = time.time()
    audio_features = whisper_encoding.get_audio_features(mel)
    languages, language_probs = whisper_decoding.detect_language(audio_features)
    tokens, sum_logprobs, no_speech_probs = whisper_decoding.main_loop(audio_features)
    result = whisper_decoding.post_process(tokens, sum_logprobs, no_speech_probs, audio_features, languages)
    print("transcribe time " + str(time.time() - begin_time))
    result = result[0]
    print(result.text)
    return result


if __name__ == '__main__':
    args = parse_arguments()
    generate(**vars(args))
