"""WER of the engines on a LibriSpeech-style directory: mirror of W/summarize.py (same flags, same text
pipeline: strip [.,!?] + upper-case, EnglishTextNormalizer on both sides, corpus WER), with the parts the
MI355X boxes lack replaced: FLAC through the library's own decoder instead of ffmpeg, the log-mel on the
GPU (wm_log_mel), jiwer.wer by wer.py.

Differences from the reference, all opt-in or additive:
  * utterances are decoded `--batch_size` at a time (the reference is batch 1) and, under torch.distributed,
    sharded over the ranks (dp.py); the WER is computed on rank 0 over the gathered hypotheses;
  * the data set may be the nested LibriSpeech tree (speaker/chapter/*.flac + *.trans.txt, what
    W/summarize.py:112-118 walks) or a flat directory such as W/LibriSpeech/valid-clean;
  * `--test_torch` needs a PyTorch Whisper with `.encoder` / `.logits` built from `--checkpoint_file`
    (openai-whisper's `whisper.model.Whisper`); without that package the flag raises.
"""
import argparse
import logging
import re
import time
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch

from decoding import WhisperDecoding
from encoding import WhisperEncoding
from normalizers import EnglishTextNormalizer
from wer import wer as word_error_rate
from whisper_utils import N_SAMPLES, load_audio, log_mel_spectrogram, log_mel_spectrogram_device, pad_or_trim

logger = logging.getLogger("whisper_mi355.summarize")
AUDIO_SUFFIXES = (".flac", ".wav", ".npy")


def load_dataset(dataset_dir) -> Tuple[List[Path], List[str]]:
    """One directory: the transcript file (`*.txt`, lines `<utterance-id> <TEXT>`) and the audio files.
    Returns (audio files sorted by name, reference texts in transcript order) — W/summarize.py:56-71."""
    label_file, audio_files = None, []
    for f in Path(dataset_dir).iterdir():
        if f.name.endswith("txt"):
            label_file = f
        elif f.suffix in AUDIO_SUFFIXES:
            audio_files.append(f)
    references = []
    if label_file is not None:
        with open(label_file) as fh:
            references = [line.split(" ", 1)[1].replace("\n", "") for line in fh if " " in line]
    return sorted(audio_files), references


def discover(dataset_dir) -> List[Tuple[Path, str]]:
    """(audio file, reference text) pairs of every directory under `dataset_dir` that holds a transcript."""
    root = Path(dataset_dir)
    leaves = sorted({p.parent for p in root.rglob("*.txt")} | ({root} if any(root.glob("*.txt")) else set()))
    pairs: List[Tuple[Path, str]] = []
    for leaf in leaves:
        audio_files, references = load_dataset(leaf)
        if len(audio_files) != len(references):
            # match by utterance id when the directory is not one-line-per-file
            by_id = {}
            for f in leaf.iterdir():
                if f.name.endswith("txt"):
                    with open(f) as fh:
                        by_id.update(dict(line.rstrip("\n").split(" ", 1) for line in fh if " " in line))
            pairs += [(a, by_id[a.stem]) for a in audio_files if a.stem in by_id]
        else:
            pairs += list(zip(audio_files, references))
    return pairs


# bytes per second of audio by container, for clips whose header does not give a duration: FLAC of 16 kHz mono speech
# compresses to roughly half of PCM16 (LibriSpeech: ~ 15-18 kB/s), wav is PCM16, a log-mel .npy holds 80 x 100 fp32 per second
_BYTES_PER_SECOND = {".flac": 16000.0, ".wav": 32000.0, ".npy": 32000.0}
_warned_fallback = [False]


def clip_seconds(audio_file) -> float:
    """Duration of a clip in SECONDS without decoding it: FLAC STREAMINFO (the fields wm_flac_info returns, read from the first
    42 bytes), the wav header, the .npy header (frames at the front end's 100 per second).  A header that gives no duration
    (a FLAC with total_samples = 0, a container this reader does not know) falls back to file size / a bytes-per-second estimate
    for the suffix -- still seconds, so such clips sort among the others instead of behind every one of them -- and says so once."""
    path = Path(audio_file)
    try:
        if path.suffix == ".flac":
            with open(path, "rb") as fh:
                head = fh.read(42)                          # "fLaC", block header, STREAMINFO (always the first block, RFC 9639)
            if len(head) == 42 and head[:4] == b"fLaC" and (head[4] & 0x7f) == 0:
                s = head[8:]
                rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4)
                total = ((s[13] & 15) << 32) | (s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17]
                if rate > 0 and total > 0:
                    return total / float(rate)
        elif path.suffix == ".wav":
            import wave
            with wave.open(str(path), "rb") as w:
                return w.getnframes() / float(w.getframerate())
        elif path.suffix == ".npy":
            arr = np.load(path, mmap_mode="r")
            # a log-mel [n_mels, frames] (100 frames per second, W/whisper_utils.py:99-146) or raw 16 kHz samples
            return arr.shape[-1] / (100.0 if arr.ndim >= 2 else 16000.0)
    except Exception:       # noqa: BLE001 -- ordering is an optimisation; any unreadable header falls back to the size estimate
        pass
    if not _warned_fallback[0]:
        _warned_fallback[0] = True
        logging.getLogger(__name__).warning("summarize: no duration in the header of %s (and possibly others): ordering by size / %s bytes per second",
                                            path, _BYTES_PER_SECOND.get(path.suffix, 16000.0))
    return float(path.stat().st_size) / _BYTES_PER_SECOND.get(path.suffix, 16000.0)


def plan_batches(pairs, batch_size: int, rank: int = 0, world: int = 1, deal=None):
    """The batches of this rank, clips ordered by duration (length-sorted chunks balance the decode lengths).
    The sorted list is cut into batches of `batch_size` and the batches are dealt round-robin over the ranks: inside a
    batch the clips are of similar length -- its utterance groups finish together and finished rows stop costing cross-K/V
    bandwidth early (per-row completion in WhisperDecoding.main_loop) -- while every rank gets short and long batches alike
    (a contiguous slice of the sorted list would hand rank 0 all the short clips).  `deal(n) -> permutation` (optional:
    WhisperDecoding.balanced_order) arranges each batch so that the decoder's utterance groups get short and long clips alike."""
    order = sorted(range(len(pairs)), key=lambda i: (clip_seconds(pairs[i][0]), str(pairs[i][0])))
    batches = [[pairs[i] for i in order[k:k + batch_size]] for k in range(0, len(order), batch_size)]
    if deal is not None:
        batches = [[b[j] for j in deal(len(b))] for b in batches]
    return batches[rank::world]


def clean_hypothesis(text: str) -> str:
    """What the reference does to the engine output before normalising (W/summarize.py:125-127)."""
    punctuations = re.findall(r"[.,!?]", text)
    return text.translate(str.maketrans({p: "" for p in punctuations})).upper()


def mel_batch(audio_list: List[np.ndarray], device) -> torch.Tensor:
    """fp16 [B, 80, 3000] on `device`: pad_or_trim each clip, STFT + mel projection on the GPU."""
    padded = np.stack([pad_or_trim(a) for a in audio_list]).astype(np.float32)
    if torch.device(device).type == "cuda":
        return log_mel_spectrogram_device(torch.from_numpy(padded).to(device), dtype=torch.float16)
    return torch.stack([log_mel_spectrogram(torch.from_numpy(p)) for p in padded]).half()


def eval_engines(whisper_encoding, whisper_decoding, mel) -> list:
    audio_features = whisper_encoding.get_audio_features(mel)
    languages, _ = whisper_decoding.detect_language(audio_features)
    tokens, sum_logprobs, no_speech_probs = whisper_decoding.main_loop(audio_features)
    return whisper_decoding.post_process(tokens, sum_logprobs, no_speech_probs, audio_features, languages)


def eval_engines_stream(whisper_encoding, whisper_decoding, mels, cu_budget: Optional[int] = None):
    """eval_engines over a sequence of mel batches, software-pipelined: while the decode loop of batch n streams its
    cross-attention K/V, the encoder of batch n + 1 runs beside it on `cu_budget` CUs (WhisperEncoding.prefetch).
    Yields one result list per batch, in order; the results are the ones eval_engines returns batch by batch."""
    from encoding import DEFAULT_SHARED_CU_BUDGET
    budget = DEFAULT_SHARED_CU_BUDGET if cu_budget is None else cu_budget
    it = iter(mels)
    mel, pending = next(it, None), False
    while mel is not None:
        audio_features = whisper_encoding.collect() if pending else whisper_encoding.get_audio_features_async(mel)
        languages, _ = whisper_decoding.detect_language(audio_features)
        mel = next(it, None)
        pending = mel is not None and budget > 0
        if pending:
            whisper_encoding.prefetch(mel, budget)
        tokens, sum_logprobs, no_speech_probs = whisper_decoding.main_loop(audio_features)
        whisper_encoding.loop_ended()          # the pass in flight gives back its CU budget once the GPU has finished this loop
        yield whisper_decoding.post_process(tokens, sum_logprobs, no_speech_probs, audio_features, languages)


def eval_torch(whisper_encoding, whisper_decoding, mel, model) -> list:
    audio_features = whisper_encoding.torch_get_audio_features(model, mel)
    languages, _ = whisper_decoding.torch_detect_language(model, audio_features)
    tokens, sum_logprobs, no_speech_probs = whisper_decoding.torch_main_loop(model, audio_features)
    return whisper_decoding.post_process(tokens, sum_logprobs, no_speech_probs, audio_features, languages)


def load_torch_model(checkpoint_file: str, device, factory: Optional[str] = None):
    """The model of the PyTorch comparison path (W/summarize.py:78-84 builds the reference's in-tree `Whisper` from the checkpoint).
    Default: this package's own PyTorch Whisper (torch_model.py) over the same checkpoint.  `factory` = "module:callable" names any
    other implementation: `callable(checkpoint_file, device)` must return an object with `.encoder(mel)`, `.logits(tokens, xa)`,
    `.decoder(tokens, xa, kv_cache=)` and `.install_kv_cache_hooks()` -- what eval_torch drives (e.g. openai-whisper's `Whisper`)."""
    if factory:
        import importlib
        mod_name, _, fn_name = factory.partition(":")
        fn = getattr(importlib.import_module(mod_name), fn_name or "load_model")
        model = fn(checkpoint_file, device)
    else:
        import torch_model
        model = torch_model.load_model(checkpoint_file, device)
    for need in ("encoder", "logits", "decoder", "install_kv_cache_hooks"):
        if not hasattr(model, need):
            raise RuntimeError(f"--torch_model {factory}: the object it returned has no `{need}`")
    return model


def score(hypotheses: List[str], references: List[str], normalizer=None) -> float:
    normalizer = normalizer or EnglishTextNormalizer()
    return word_error_rate([normalizer(t) for t in references], [normalizer(t) for t in hypotheses])


def transcribe_dataset_stream(pairs, stream_evaluate, batch_size: int, device) -> Tuple[List[str], List[str], float]:
    """transcribe_dataset for a pipelined evaluator: `stream_evaluate(iterator of mel batches)` yields one result list per
    batch (eval_engines_stream).  The batches are formed lazily, so the audio of batch n + 1 is read and its log-mel computed
    while batch n decodes; the seconds returned are the wall time of the whole loop (file reading included)."""
    import collections
    hyps, refs, waiting = [], [], collections.deque()

    def batches():
        audio, texts = [], []
        for audio_file, ref in pairs:
            a = load_audio(str(audio_file))
            if a.shape[-1] > N_SAMPLES:
                continue
            audio.append(a)
            texts.append(ref)
            if len(audio) == batch_size:
                waiting.append(list(texts))
                yield mel_batch(audio, device)
                audio, texts = [], []
        if audio:
            waiting.append(list(texts))
            yield mel_batch(audio, device)

    t0 = time.time()
    for results in stream_evaluate(batches()):
        for ref, res in zip(waiting.popleft(), results):
            hyps.append(clean_hypothesis(res.text))
            refs.append(ref)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    return hyps, refs, time.time() - t0


def transcribe_dataset(pairs, evaluate, batch_size: int, device) -> Tuple[List[str], List[str], float]:
    """Run `evaluate(mel) -> [DecodingResult]` over the clips that fit 30 s; returns (hypotheses, references,
    seconds spent in `evaluate`)."""
    hyps, refs, batch_audio, batch_refs = [], [], [], []
    elapsed = 0.0

    def flush():
        nonlocal elapsed
        if not batch_audio:
            return
        mel = mel_batch(batch_audio, device)
        torch.cuda.synchronize() if mel.is_cuda else None
        t0 = time.time()
        results = evaluate(mel)
        torch.cuda.synchronize() if mel.is_cuda else None
        elapsed += time.time() - t0
        for ref, res in zip(batch_refs, results):
            hyps.append(clean_hypothesis(res.text))
            refs.append(ref)
            logger.info("---------------------------------------------------------")
            logger.info(f"\n Reference : {ref}")
            logger.info(f"\n Output : {hyps[-1]}")
        batch_audio.clear()
        batch_refs.clear()

    for audio_file, ref in pairs:
        audio = load_audio(str(audio_file))
        if audio.shape[-1] > N_SAMPLES:          # the reference skips clips longer than one window (summarize.py:114-115)
            continue
        batch_audio.append(audio)
        batch_refs.append(ref)
        if len(batch_audio) == batch_size:
            flush()
    flush()
    return hyps, refs, elapsed


def rank_share(pairs, batch_size: int, rank: int, world: int, deal=None, keep_order: bool = False) -> list:
    """The (audio file, reference) pairs THIS rank transcribes, in the order it takes them (W/summarize.py:72-181 is one
    process over the whole list).  Default: batches of similar-length clips dealt round-robin over the ranks (plan_batches) --
    the last batch may be ragged and some ranks may get one batch fewer; `keep_order` (or batch size 1): a contiguous slice of the
    directory order (dp.shard_bounds).  Every pair lands on exactly one rank; the WER does not depend on the order."""
    import dp
    if keep_order or batch_size <= 1:
        lo, hi = dp.shard_bounds(len(pairs), rank, world)
        return list(pairs[lo:hi])
    return [pair for batch in plan_batches(pairs, batch_size, rank, world, deal) for pair in batch]


def gather_transcripts(hyps: List[str], refs: List[str], seconds: float) -> Tuple[List[str], List[str], float]:
    """Every rank's hypotheses and references on every rank (rank order; pairs stay aligned), and the slowest rank's seconds:
    the job's only exchange besides the model weights each rank loads for itself.  Without a process group: the arguments."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return hyps, refs, seconds
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, (hyps, refs, seconds))
    return [h for g in gathered for h in g[0]], [r for g in gathered for r in g[1]], max(g[2] for g in gathered)


def main(args) -> Optional[dict]:
    logging.basicConfig(level=getattr(logging, args.log_level.upper(), logging.INFO))
    import torch.distributed as dist
    import dp
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    device = torch.device(args.device) if args.device else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    if args.test_trt_llm and device.type != "cuda":
        raise RuntimeError("--test_trt_llm runs the HIP engines: it needs a GPU (only --test_torch runs on the host)")
    engine_dir = Path(args.engine_dir)
    only_torch = not args.test_trt_llm                 # the PyTorch path alone needs the engine directory's configuration, not its engines
    whisper_encoding = WhisperEncoding(engine_dir, only_torch=only_torch)
    whisper_decoding = WhisperDecoding(engine_dir, only_torch=only_torch, vocab_path=args.vocab)
    if args.sample_len:
        whisper_decoding.sample_len = args.sample_len
    pairs = discover(args.dataset_dir)
    mine = rank_share(pairs, args.batch_size, rank, world, whisper_decoding.balanced_order, args.no_sort_by_duration)
    report = {}
    runs = []
    if args.test_torch:
        model = load_torch_model(args.checkpoint_file, device, args.torch_model)
        runs.append(("Torch", lambda mel: eval_torch(whisper_encoding, whisper_decoding, mel, model)))
    if args.test_trt_llm:
        runs.append(("whisper-mi355", lambda mel: eval_engines(whisper_encoding, whisper_decoding, mel)))
    for name, evaluate in runs:
        if name == "whisper-mi355" and args.overlap_encoder:
            hyps, refs, seconds = transcribe_dataset_stream(
                mine, lambda mels: eval_engines_stream(whisper_encoding, whisper_decoding, mels), args.batch_size, device)
        else:
            hyps, refs, seconds = transcribe_dataset(mine, evaluate, args.batch_size, device)
        hyps, refs, seconds = gather_transcripts(hyps, refs, seconds)
        if rank == 0:
            value = score(hyps, refs)
            logger.info(f"{name} (total latency: {seconds} sec)")
            logger.info(f"{name} beam 0 result")
            logger.info(f"\nWER: {value * 100:.2f} %")
            report[name] = dict(wer=value, seconds=seconds, utterances=len(hyps), hypotheses=list(hyps))
    return report if rank == 0 else None


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--test_torch', action='store_true')
    parser.add_argument('--test_trt_llm', action='store_true', help='evaluate the engines (flag name kept from the reference)')
    parser.add_argument('--data_type', type=str, choices=['fp16'], default='fp16')
    parser.add_argument('--log_level', type=str, default='info')
    parser.add_argument('--engine_dir', type=str, default='whisper_outputs')
    parser.add_argument('--dataset_dir', type=str, default='./LibriSpeech/test-clean')
    parser.add_argument('--checkpoint_file', type=str, default='./large-v2.pt')
    parser.add_argument('--batch_size', type=int, default=32, help='utterances decoded together (the reference: 1)')
    parser.add_argument('--vocab', type=str, default=None, help='path to multilingual.tiktoken / gpt2.tiktoken')
    parser.add_argument('--no_sort_by_duration', action='store_true',
                        help='keep the directory order instead of batching clips of similar duration (plan_batches)')
    parser.add_argument('--torch_model', type=str, default=None,
                        help='--test_torch: "module:callable" returning the PyTorch model for (checkpoint_file, device); default: torch_model.load_model')
    parser.add_argument('--device', type=str, default=None, help='default: the current GPU (cpu serves --test_torch alone)')
    parser.add_argument('--sample_len', type=int, default=None, help='tokens sampled per utterance at most (default: n_text_ctx // 2)')
    parser.add_argument('--overlap_encoder', action='store_true',
                        help='run the encoder of the next batch beside the decode loop of the current one (eval_engines_stream)')
    return parser.parse_args(argv)


if __name__ == '__main__':
    main(parse_arguments())
