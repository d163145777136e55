import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd")]
import torch, synthetic
from pathlib import Path
from decoding import WhisperDecoding
from encoding import WhisperEncoding
eng = Path("/tmp/wm_bench_engines/large-v2-int8-seed0")
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
dec.sample_len = 12
B = int(sys.argv[1]); dec.micro_batches = int(sys.argv[2]); dec.use_graphs = bool(int(sys.argv[3]))
mel = synthetic.synthetic_mel(B, 3000, 80, 1234).cuda()
xa = enc.get_audio_features_async(mel)
dec.detect_language(xa)
dec.main_loop(xa, ignore_eot=True)
torch.cuda.synchronize()
dec.main_loop(xa, ignore_eot=True)
torch.cuda.synchronize()
