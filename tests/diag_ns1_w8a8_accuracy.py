"""ns1 experiment, accuracy half (CPU, the oracle): what int8 MFMA would change numerically.

BASELINE.json's north_star asks for "int8 MFMA ... for the dense weight-only GEMMs".  The reference's weight-only contract is
W8A16: int8 weights expanded to fp16, fp16 activations, fp32 accumulation (weightOnlyMatrixVectorMultiplication.cu:44-53,
fpA_intB_gemm_template.h:47-140).  v_mfma_i32_16x16x64_i8 needs int8 ACTIVATIONS too, i.e. W8A8 with a dynamic per-token scale
(SmoothQuant-style, which the reference implements for GPT only).  This script measures, with the CPU oracle, what that would
do to the M = 1500 x batch GEMMs it could serve -- the encoder blocks' Linears and the cross-K/V projection:

    x_q = clip(rne(x / s_t), -127, 127),  s_t = max|x_t| / 127 per token row;   y = (x_q . w_q^T) * s_t * s_c  (+ bias) -> fp16

against the W8A16 oracle (same int8 weights), on the tiny.en-shaped 2 + 2 layer model of tests/golden (384 wide, 1500 audio
positions) and on the micro model.  Prints max |d| of the encoder output, of the cross K/V and of teacher-forced logits, and
greedy-id agreement.   usage: python tests/diag_ns1_w8a8_accuracy.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle.whisper_oracle import Dims, MICRO, OracleConfig, OracleModel, greedy_reference_run, synthetic_mel, synthetic_state_dict, _r


class W8A8Model(OracleModel):
    """OracleModel whose encoder-block and cross-K/V Linears quantise their INPUT rows to int8 with a per-token scale."""
    def _linear(self, x, wkey, bkey=None):
        big_m = wkey.startswith("encoder.blocks.") or ".cross_attn.key." in wkey or ".cross_attn.value." in wkey
        if not (big_m and wkey in self.q):
            return super()._linear(x, wkey, bkey)
        q, s = self.q[wkey]                                              # int8 [out, in], fp16 [out]
        st = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-8) / 127.0
        xq = torch.clamp(torch.round(x / st), -127, 127)
        acc = (xq.double() @ torch.from_numpy(q.astype(np.float64)).t()).float()          # exact integer sums
        y = acc * st * torch.from_numpy(s.astype(np.float32))
        if bkey is not None:
            y = y + self.p[bkey]
        return _r(y, self.cfg.act)


def compare(name, dims, seed, batch, n_steps):
    sd = synthetic_state_dict(dims, seed)
    mel = synthetic_mel(batch, 2 * dims.n_audio_ctx, dims.n_mels, 4242)
    prompt = [dims.n_vocab - 1607, dims.n_vocab - 1606, dims.n_vocab - 1506] if dims.n_vocab > 50000 else [5, 17, 900]
    with torch.no_grad():
        ref = greedy_reference_run(OracleModel(dims, sd, OracleConfig(act="float16", weight_only=True)), mel, prompt, n_steps)
        m8 = W8A8Model(dims, sd, OracleConfig(act="float16", weight_only=True))
        xa = m8.encoder(mel)
        ckv = m8.cross_kv(xa)
        kv, worst, agree, safe = None, 0.0, 0, 0
        cur = torch.tensor([prompt] * batch)
        for s in range(n_steps):                                         # teacher-forced with the W8A16 oracle's ids
            logits, kv = m8.decoder(cur, ckv, kv)
            worst = max(worst, float((logits[:, -1] - ref["logits"][s][:, -1]).abs().max()))
            ok = logits[:, -1].argmax(-1) == ref["ids"][:, s]
            m = ref["margins"][:, s] > 0.06
            agree += int((ok & m).sum()); safe += int(m.sum())
            cur = ref["ids"][:, s:s + 1]
    d_xa = float((xa - ref["xa"]).abs().max())
    d_ckv = max(float((a - b).abs().max()) for a, b in zip(ckv, ref["cross_kv"]))
    print(f"{name}: W8A8 (per-token dynamic) vs W8A16 oracle: max|d xa| = {d_xa:.4f} (|xa| max {float(ref['xa'].abs().max()):.2f}), "
          f"max|d cross K/V| = {d_ckv:.4f}, max|d logits| = {worst:.4f} over {n_steps} steps, greedy ids {agree}/{safe} on margins > 0.06")


if __name__ == "__main__":
    torch.set_num_threads(8)
    compare("micro (128 wide, 2+2 layers)", MICRO, 7, 2, 6)
    compare("tiny.en shape (384 wide, 1500 positions, 2+2 layers)", Dims(80, 1500, 384, 6, 2, 51864, 448, 384, 6, 2), 33, 1, 5)
    compare("large-v2 width (1280 wide, 1500 positions, 1+1 layers)", Dims(80, 1500, 1280, 20, 1, 51865, 448, 1280, 20, 1), 12, 1, 4)
