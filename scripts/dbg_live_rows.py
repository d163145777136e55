"""Debug: one decode step with a live-row list, chain modes 0 / 2, batch B: which rows / layers differ from the full step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "eddie-wang-hackathon2023_amd"), os.path.join(ROOT, "tests")]
import tempfile
from pathlib import Path
import torch
import native, synthetic
from synthetic import synthetic_mel
from test_gpu_model import build_engine
from encoding import WhisperEncoding
from decoding import WhisperDecoding
lib = native.load_library()
lib.wm_set_small_batch_rows(8)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
from types import SimpleNamespace
dims = SimpleNamespace(**synthetic.DIMS["micro"])
eng = build_engine(Path(tempfile.mkdtemp()), "micro", 7, True, True, [0.05, 0.06])
enc, dec = WhisperEncoding(eng), WhisperDecoding(eng)
mel = synthetic_mel(batch, 2 * dims.n_audio_ctx, dims.n_mels, 5).cuda()
cross = dec.xa2cross_key_value(enc.get_audio_features(mel))
cap, H, V = dims.n_text_ctx, dims.n_text_head, dims.n_vocab
sess, pos = dec.decoder_session, dec.positional_embedding
g = torch.Generator().manual_seed(batch)
toks = torch.randint(0, V, (batch, 4), generator=g).to(torch.int32).cuda()
s = torch.cuda.current_stream().cuda_stream
def run(live):
    kv = [torch.zeros((batch, 2, H, cap, 64), dtype=torch.int8, device="cuda") for _ in range(dims.n_text_layer)]
    lg = torch.zeros((batch, 3, V), dtype=torch.float16, device="cuda")
    sess.decoder_step(toks[:, :3].contiguous(), pos[0:3], cross, None, cap, kv, cap, lg, 0, s)
    lg1 = torch.full((batch, 1, V), 123.0, dtype=torch.float16, device="cuda")
    sess.decoder_step(toks[:, 3:4].contiguous(), pos[3:4], cross, kv, cap, kv, cap, lg1, 3, s, live_rows=live)
    torch.cuda.synchronize()
    return lg1, kv
rows = [r for r in range(batch) if r % 3 != 1]
live = torch.tensor([len(rows)] + rows + [0] * (batch - len(rows)), dtype=torch.int32, device="cuda")
res = {}
for mode in (0, 2):
    lib.wm_set_decode_chain(mode)
    before = native.chain_status()["launches"]
    res[mode, "full"] = run(None)
    res[mode, "live"] = run(live)
    print("mode", mode, "chain launches", native.chain_status()["launches"] - before, native.chain_status()["reason"])
for a, b in (((0, "full"), (2, "full")), ((0, "full"), (0, "live")), ((2, "full"), (2, "live")), ((0, "live"), (2, "live"))):
    la, ka = res[a]; lb, kb = res[b]
    print(a, "vs", b, "logits equal per row:", [bool(torch.equal(la[r], lb[r])) for r in range(batch)],
          "max diff live rows", float((la[rows].float() - lb[rows].float()).abs().max()),
          "cache equal per layer/row:", [[bool(torch.equal(x[r], y[r])) for r in range(batch)] for x, y in zip(ka, kb)])
