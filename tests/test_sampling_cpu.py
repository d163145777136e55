"""The sampling path of the decode loop (temperature > 0, best_of, MaximumLikelihoodRanker with and
without a length penalty) held to goldens the REFERENCE's own classes produced (tests/golden/sampling.npz, written by
oracle/gen_golden.py: W/decoding.py GreedyDecoder :274-300, MaximumLikelihoodRanker :92-115, main_loop :785-821 and
post_process :827-878 run as they stand, on the seeded logits of oracle.decoding_rules.sampling_logits).

These are host-side rules (torch on CPU tensors, torch's CPU generator for the Categorical draw), so the whole row
is checked here without a GPU: the product's `decoding.GreedyDecoder`, `MaximumLikelihoodRanker`, the candidate
grouping of `main_loop_reference` (what `main_loop` routes to whenever temperature != 0 or best_of > 1) and
`post_process` must reproduce the reference's tokens, log-probabilities, ranks and selections exactly."""
import os

import numpy as np
import pytest
import torch

import build as B
import synthetic
from decoding import DecodingOptions, GreedyDecoder, MaximumLikelihoodRanker, WhisperDecoding
from oracle import decoding_rules as DR

IDS = DR.MULTILINGUAL


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "sampling.npz"))


@pytest.fixture(scope="module")
def engine_dir(tmp_path_factory):
    out = tmp_path_factory.mktemp("sampling") / "eng"
    args = B.parse_arguments(["--output_dir", str(out), "--log_level", "error"])
    B.build_from_checkpoint(synthetic.synthetic_checkpoint("micro-fullvocab", 3), args)
    # host rules only (no engine runs here): give the config large-v2's 1500 audio positions, i.e. 0.02 s per timestamp
    # and max_initial_timestamp_index = 50, which is what the reference's constructor derives (W/decoding.py:339-348)
    import json
    cfg = json.load(open(out / "decoder_config.json"))
    cfg["builder_config"]["num_audio_ctx"] = 1500
    json.dump(cfg, open(out / "decoder_config.json", "w"))
    return out


def test_greedy_decoder_with_temperature_matches_reference(fx):
    """Three consecutive updates of ONE decoder at temperature 0.7 (the generator state carries over as in a loop),
    one row already at EOT: next tokens, running log-probability sums, completion flag, finalize."""
    greedy = GreedyDecoder(0.7, IDS.eot)
    torch.manual_seed(1234)
    tokens = torch.tensor([[IDS.sot, IDS.lang0, IDS.transcribe]] * 5)
    tokens[3, -1] = IDS.eot
    sum_lp = torch.zeros(5)
    for step in range(3):
        lg = torch.from_numpy(DR.sampling_logits(step + 10, 5, 1)[:, 0].copy())
        tokens, done = greedy.update(tokens, lg, sum_lp)
        assert np.array_equal(tokens[:, -1].numpy(), fx[f"upd{step}_next"]), step
        assert np.array_equal(sum_lp.numpy(), fx[f"upd{step}_sumlp"]), step          # same torch ops, same order: bit-equal
        assert bool(done) == bool(fx[f"upd{step}_done"])
    assert int(tokens[3, -1]) == IDS.eot and float(sum_lp[3]) == 0.0                 # finished rows stay at EOT, add nothing
    ftok, flp = greedy.finalize(tokens.reshape(1, 5, -1), sum_lp.reshape(1, 5))
    assert np.array_equal(ftok.numpy(), fx["upd_final_tokens"]) and np.array_equal(np.array(flp), fx["upd_final_sumlp"])


@pytest.mark.parametrize("tag,penalty", [("none", None), ("0p6", 0.6), ("1p0", 1.0)])
def test_maximum_likelihood_ranker_matches_reference(fx, tag, penalty):
    lengths, sums = fx["rank_lengths"].tolist(), fx["rank_sumlp"].tolist()
    toks = [[torch.zeros(n, dtype=torch.long) for n in g] for g in lengths]
    got = MaximumLikelihoodRanker(penalty).rank(toks, sums)
    assert [int(i) for i in got] == fx[f"rank_{tag}"].tolist()
    assert DR.rank_max_likelihood(lengths, sums, penalty) == fx[f"rank_{tag}"].tolist()       # the oracle's restatement too
    if tag != "none":
        assert fx[f"rank_{tag}"].tolist() != fx["rank_none"].tolist()                         # the cases do tell the rules apart


@pytest.mark.parametrize("tag,penalty", [("none", None), ("0p6", 0.6)])
def test_sampling_loop_and_post_process_match_reference(fx, engine_dir, tag, penalty, monkeypatch):
    """best_of = 3 at temperature 0.7 through the product's loop: candidate rows are grouped per utterance
    (repeat_interleave), every step is one `decode()` of all candidates, rows that sampled EOT stay there, the loop
    stops when all have, post_process slices between sample_begin and the first EOT and ranks per utterance."""
    n_audio, n_group = 2, 3
    dec = WhisperDecoding(engine_dir, only_torch=True,
                          options=DecodingOptions(temperature=0.7, best_of=n_group, sample_len=12, length_penalty=penalty))
    assert dec.n_group == n_group and dec.sample_begin == 3 and dec.sot_index == 0 and dec.max_initial_timestamp_index == 50
    assert sorted(dec.logit_filters[1].suppress_tokens) == fx["suppress"].tolist()
    dec.tokenizer.decode = lambda t: " ".join(str(int(x)) for x in t)          # the goldens carry ids, not text
    calls = []

    def decode(x, cross, past=None):
        calls.append(tuple(x.shape))
        return torch.from_numpy(DR.sampling_logits(len(calls) - 1, x.shape[0], x.shape[1])), None
    monkeypatch.setattr(dec, "decode", decode)
    monkeypatch.setattr(dec, "xa2cross_key_value", lambda xa: None)
    dec.tokens = torch.tensor([dec.initial_tokens]).repeat(n_audio, 1)
    xa = torch.zeros(n_audio, 1, 1)                                          # one feature row per utterance (ours), candidates share it
    torch.manual_seed(99)
    tokens, sum_lp, nsp = dec.main_loop(xa)                                  # temperature != 0 -> main_loop_reference
    assert calls == [tuple(c) for c in fx[f"loop_{tag}_calls"].tolist()]
    assert np.array_equal(tokens.numpy(), fx[f"loop_{tag}_tokens"])
    assert np.array_equal(sum_lp.numpy(), fx[f"loop_{tag}_sumlp"])
    assert np.allclose(np.array(nsp), fx[f"loop_{tag}_nsp"], rtol=0, atol=0)
    res = dec.post_process(tokens, sum_lp, nsp, xa, ["en"] * n_audio)
    assert [" ".join(map(str, r.tokens)) for r in res] == fx[f"loop_{tag}_selected_tokens"].tolist()
    assert np.array_equal(np.array([r.avg_logprob for r in res]), fx[f"loop_{tag}_avg_logprob"])
    assert np.array_equal(np.array([r.no_speech_prob for r in res]), fx[f"loop_{tag}_nsp_selected"])
    assert all(r.temperature == 0.7 for r in res)
    # the candidates of one utterance did end at different lengths (the ranker's length term had something to weigh)
    lens = [(row[3:] != IDS.eot).sum() for row in fx[f"loop_{tag}_tokens"]]
    assert len(set(lens[:3])) > 1 and len(set(lens[3:])) > 1
