"""The data-parallel layer (dp.py) on CPU: world_size 2 over gloo.  Utterances are sharded with no
data-path collective; the only exchanges are the mel scatter and the result gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dp


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 8, 9, 100):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        feat = (4, 6)
        mels = torch.arange(n_items * 24, dtype=torch.float32).reshape(n_items, *feat) if rank == 0 else None
        mine = dp.scatter_utterances(mels, n_items, feat, torch.float32, "cpu")
        lo, hi = dp.shard_bounds(n_items, rank, world)
        want = torch.arange(n_items * 24, dtype=torch.float32).reshape(n_items, *feat)[lo:hi]
        assert torch.equal(mine, want), (rank, mine.shape)
        # each rank "decodes": token row = utterance id, ragged widths per rank
        width = 5 + rank
        toks = (torch.arange(lo, hi)[:, None] * 10 + torch.arange(width)[None, :]).to(torch.int64)
        lps = torch.arange(lo, hi, dtype=torch.float32) * -1.5
        out = dp.gather_results(toks, lps, n_items, 8, pad_value=-1)
        t = dp.max_over_ranks(float(rank + 1), "cpu")
        assert t == float(world)
        if rank == 0:
            all_t, all_lp = out
            assert all_t.shape == (n_items, 8) and all_lp.shape == (n_items,)
            for r in range(world):
                l2, h2 = dp.shard_bounds(n_items, r, world)
                w = 5 + r
                assert torch.equal(all_t[l2:h2, :w], (torch.arange(l2, h2)[:, None] * 10 + torch.arange(w)[None, :]))
                assert (all_t[l2:h2, w:] == -1).all()
            assert torch.equal(all_lp, torch.arange(n_items, dtype=torch.float32) * -1.5)
        else:
            assert out is None
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8])
def test_scatter_gather_world2_gloo(n_items):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n_items, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_single_process_passthrough():
    mels = torch.randn(3, 2, 5)
    assert torch.equal(dp.scatter_utterances(mels, 3, (2, 5), torch.float32, "cpu"), mels)
    toks = torch.tensor([[1, 2, 3], [4, 5, 6]])
    out, lp = dp.gather_results(toks, torch.tensor([0.5, 1.5]), 2, 5, pad_value=9)
    assert out.tolist() == [[1, 2, 3, 9, 9], [4, 5, 6, 9, 9]] and lp.tolist() == [0.5, 1.5]
    assert dp.max_over_ranks(3.25, "cpu") == 3.25


def test_bench_launches_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher starts N fresh ranks through torch.distributed.run (never an exec of
    a GPU-initialised process), relays rank 0's JSON line and its exit code; with too few GPUs it refuses loudly
    instead of measuring one GPU and calling it N."""
    import argparse
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        line = '{"metric": "decode tokens/s", "value": 1.0, "n_gpus": 4}'
        return subprocess.CompletedProcess(cmd, 0, stdout="noise from a rank\n" + line + "\n")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    assert bench.launch_ranks(argparse.Namespace(gpus=4)) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "decode tokens/s", "value": 1.0, "n_gpus": 4}' and "noise from a rank" in out.err
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.launch_ranks(argparse.Namespace(gpus=4)) == 2
    assert "shows 1 GPU" in capsys.readouterr().err


def _summarize_worker(rank, world, port, tmp, ret):
    """summarize.py's rank-sharded path without a GPU: the planning (rank_share), the per-rank loop (transcribe_dataset with a
    stand-in evaluator and front end) and the gather of the transcripts; 11 clips in batches of 4 -> a ragged last batch and
    one rank with a batch fewer."""
    import types
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import summarize as S
        from pathlib import Path
        files = sorted(Path(tmp).glob("*.npy"))
        pairs = [(f, f.stem.upper()) for f in files]
        mine = S.rank_share(pairs, 4, rank, world)
        sizes = []

        def fake_eval(mel):                                     # "transcribes" a clip to its first sample (set per file below)
            sizes.append(mel.shape[0])
            return [types.SimpleNamespace(text=f"clip{int(v)}.") for v in mel[:, 0].tolist()]
        S.load_audio = lambda path: np.load(path)               # noqa: E731 -- the stand-in front end: samples as stored
        S.mel_batch = lambda audio, device: torch.tensor([[float(a[0])] for a in audio])      # noqa: E731
        hyps, refs, seconds = S.transcribe_dataset(mine, fake_eval, 4, "cpu")
        assert sizes == ([4, 3] if rank == 0 else [4]), (rank, sizes)       # batches 0 and 2 (ragged) on rank 0, batch 1 on rank 1
        hyps, refs, seconds = S.gather_transcripts(hyps, refs, float(rank + 1))
        assert seconds == float(world)
        assert len(hyps) == len(refs) == 11
        assert sorted(refs) == sorted(p[1] for p in pairs)                  # every clip exactly once
        assert all(h == r for h, r in zip(hyps, refs))                      # hypotheses stay aligned with their references
        assert dp.all_ranks(float(10 + rank), "cpu") == [10.0, 11.0]
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_summarize_rank_sharded_path_world2_gloo(tmp_path):
    import numpy as np
    for k, n in enumerate([16000, 4000, 52000, 9000, 30000, 2500, 41000, 12000, 7000, 60000, 21000]):
        a = np.zeros(n, dtype=np.float32)
        a[0] = k                                                # the "content" of clip k: its transcript is CLIPk
        np.save(tmp_path / f"clip{k}.npy", a)
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_summarize_worker, args=(2, port, str(tmp_path), ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
