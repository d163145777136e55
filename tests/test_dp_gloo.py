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


def _bench_main_worker(rank, world, port, out_dir, extra):
    """One rank of `bench.py --gpus 2 --stub-engine`: the environment torch.distributed.run would give it, stdout into a file."""
    import contextlib
    import io
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "5", "--decode-steps", "6", "--stub-engine"] + list(extra)
    before = sorted(os.sched_getaffinity(0))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    with open(os.path.join(out_dir, f"rank{rank}.out"), "w") as f:
        f.write(buf.getvalue())
    with open(os.path.join(out_dir, f"rank{rank}.aff"), "w") as f:
        f.write(",".join(map(str, sorted(os.sched_getaffinity(0)))) + "|" + ",".join(map(str, before)))
    assert not dist.is_initialized()          # main() tore its process group down (before rank 0's host-side epilogue)


@pytest.mark.parametrize("extra", [(), ("--scatter-inputs",), ("--encoder-cus", "0", "--length-dist", "forced")])
def test_bench_main_control_flow_world2_gloo(tmp_path, extra):
    """the WHOLE control flow of bench.main with two ranks before an 8-GPU node ever runs it -- placement of the ranks,
    per-rank input shards (or the scatter from rank 0), warm-up, the barrier-bracketed timed steps with the pipelined encoder hand-over,
    the gather of every step, max-over-ranks, the per-rank lists, the second figure's extra batches, the final barrier and the
    process group's teardown BEFORE rank 0's host-side epilogue -- on CPU over gloo with stand-in engines (bench.py --stub-engine).
    Exactly ONE JSON line, from rank 0, with n_gpus 2; the gathered rows are both ranks' rows in utterance order."""
    import json
    port = _free_port()
    mp.spawn(_bench_main_worker, args=(2, port, str(tmp_path), extra), nprocs=2, join=True)
    out0, out1 = (tmp_path / "rank0.out").read_text(), (tmp_path / "rank1.out").read_text()
    assert out1.strip() == "", "only rank 0 prints"
    lines = [l for l in out0.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "tokens/s" and d["vs_baseline"] is None and "cpu_baseline" not in d       # (the CPU leg is a one-rank leg)
    assert d["config"]["batch_per_gpu"] == 5 and "dp2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 5 * 6 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"] + 0.1     # whole-job tokens / max-over-ranks time
    pr = d["ms_per_step_per_rank"]
    assert len(pr["all"]) == 2 and pr["max"] == max(pr["all"]) and d["ms_per_step"] >= pr["max"] - 0.02
    assert d["gathered"]["rows"] == d["gathered"]["rows_expected"] == 10 and d["gathered"]["width"] == 3 + 6
    assert len(d["inputs"]["ms_per_rank"]) == 2 and d["inputs"]["bytes_per_rank"] == 5 * 80 * 3000 * 2
    assert ("scatter" in d["inputs"]["mode"]) == ("--scatter-inputs" in extra)
    # the gathered rows: every rank's own shard, rank order = utterance order (the stand-in's token rows are a function of the clip alone)
    import bench
    import dp as _dp
    want = []
    for r in range(2):
        if "--scatter-inputs" in extra:
            g = torch.Generator().manual_seed(1234)
            mel = (torch.randn((10, 80, 3000), generator=g) * 0.5).clamp_(-0.5, 1.5).half()[r * 5:(r + 1) * 5]
        else:
            g = torch.Generator().manual_seed(_dp.rank_seed(1234, r))
            mel = (torch.randn((5, 80, 3000), generator=g) * 0.5).clamp_(-0.5, 1.5).half()
        want.append(bench._StubEngines.token_rows(mel.float().mean(dim=(1, 2)), 9))
    assert d["gathered"]["token_checksum"] == int(torch.cat(want).sum().item() % (1 << 31))
    # placement: two ranks on this host were pinned to disjoint CPU sets (the GPU's NUMA node is unknown here: even slices)
    aff = [(tmp_path / f"rank{r}.aff").read_text().split("|") for r in range(2)]
    sets = [set(a[0].split(",")) for a in aff]
    n_before = len(aff[0][1].split(","))
    if n_before >= 2:
        assert d["affinity"]["pinned"] and sets[0].isdisjoint(sets[1]) and sum(d["affinity"]["cpus_per_rank"]) <= n_before
    if "--length-dist" not in extra:
        sf = d["second_figure"]
        assert sf["tokens_match_the_limits"] is True and sf["useful_tokens_per_s"] > 0 and sf["ms_per_batch_pipelined"] is not None


def test_rank_placement_plan():
    """dp.plan_rank_cpus: ranks that share their GPU's NUMA node get disjoint slices of it; an unknown node falls back to even slices."""
    node_cpus = {0: list(range(0, 48)) + list(range(96, 144)), 1: list(range(48, 96)) + list(range(144, 192))}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = range(192)
    got = [dp.plan_rank_cpus(r, 8, allowed, nodes, node_cpus)[0] for r in range(8)]
    assert all(len(g) == 24 for g in got) and len(set().union(*map(set, got))) == 192
    assert set(got[0]) <= set(node_cpus[0]) and set(got[5]) <= set(node_cpus[1])
    assert dp.plan_rank_cpus(3, 8, range(32), [None] * 8, {})[0] == [12, 13, 14, 15]
    cpus, why = dp.plan_rank_cpus(3, 8, range(16), [None] * 8, {})          # 2 CPUs per rank: below the floor -> nothing is pinned
    assert cpus == list(range(16)) and 'not pinned' in why
    few = {0: [0, 1, 2, 3], 1: [4, 5, 6, 7]}                                  # a node too small to cut: the whole node, never a sliver
    assert dp.plan_rank_cpus(1, 4, range(8), [0, 0, 1, 1], few)[0] == [0, 1, 2, 3]
    assert dp.plan_rank_cpus(0, 1, range(8), [None], {})[0] == list(range(8))
    assert dp._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert dp.rank_seed(1234, 0) == 1234 and len({dp.rank_seed(1234, r) for r in range(8)}) == 8


def test_gpu_numa_node_from_a_sysfs_tree(tmp_path, monkeypatch):
    """dp.gpu_numa_node / pin_rank_to_gpu_numa against a made-up sysfs: two CPU nodes in the KFD topology (simd_count 0), then four GPUs on
    render minors 128..131, two per NUMA node; HIP_VISIBLE_DEVICES re-maps local ranks; a GPU whose numa_node reads -1 is 'unknown'."""
    topo = tmp_path / "class/kfd/kfd/topology/nodes"
    for n, (simd, minor) in enumerate([(0, -1), (0, -1), (1024, 128), (1024, 129), (1024, 130), (1024, 131)]):
        d = topo / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 48}\nsimd_count {simd}\ndrm_render_minor {minor}\n")
    for minor, node in ((128, 0), (129, 0), (130, 1), (131, -1)):
        d = tmp_path / f"class/drm/renderD{minor}/device"
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpus in ((0, "0-3"), (1, "4-7")):
        d = tmp_path / f"devices/system/node/node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    sysfs = str(tmp_path)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert [dp.gpu_numa_node(i, sysfs) for i in range(5)] == [0, 0, 1, None, None]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert dp._visible_index(0) == 2 and dp._visible_index(1) == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    before = sorted(os.sched_getaffinity(0))
    try:
        rep = dp.pin_rank_to_gpu_numa(1, 4, sysfs)       # rank 1's GPU sits on node 0 with rank 0: the second half of node 0's CPUs that are allowed here
        allowed0 = [c for c in (0, 1, 2, 3) if c in before]
        if len(allowed0) >= 4:      # two ranks share node 0's four CPUs: too few to cut (dp.MIN_CPUS_PER_RANK) -> the whole node
            assert rep["gpu_numa_node"] == 0 and sorted(os.sched_getaffinity(0)) == [0, 1, 2, 3] and rep["pinned"] == (len(before) > 4)
        rep3 = dp.pin_rank_to_gpu_numa(3, 4, sysfs)      # numa_node -1: unknown -> an even slice of what is allowed NOW (never raises)
        assert rep3["gpu_numa_node"] is None and "unknown" in rep3["how"]
    finally:
        os.sched_setaffinity(0, before)


def test_rank_placement_plan_properties():
    """dp.plan_rank_cpus over random hosts (hypothesis): every rank gets CPUs it is allowed to use; a rank is never squeezed below
    dp.MIN_CPUS_PER_RANK unless it gets a whole node / everything; ranks that are cut apart on one node never share a CPU; a GPU on a known
    node stays on that node."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 8), st.integers(1, 4), st.integers(1, 64), st.data())
    def check(local_world, n_nodes, cpus_per_node, data):
        node_cpus = {n: list(range(n * cpus_per_node, (n + 1) * cpus_per_node)) for n in range(n_nodes)}
        allowed = sorted(data.draw(st.sets(st.integers(0, n_nodes * cpus_per_node - 1), min_size=1)))
        nodes = [data.draw(st.one_of(st.none(), st.integers(0, n_nodes - 1))) for _ in range(local_world)]
        plans = [dp.plan_rank_cpus(r, local_world, allowed, nodes, node_cpus) for r in range(local_world)]
        for r, (cpus, why) in enumerate(plans):
            assert cpus and set(cpus) <= set(allowed), (r, why)
            on_node = nodes[r] is not None and any(c in set(allowed) for c in node_cpus[nodes[r]])
            if on_node:
                assert set(cpus) <= set(node_cpus[nodes[r]]), (r, why)
            if len(cpus) < dp.MIN_CPUS_PER_RANK:        # only ever the whole node's allowed CPUs or everything allowed
                whole = [c for c in node_cpus[nodes[r]] if c in set(allowed)] if on_node else allowed
                assert cpus == whole, (r, why)
        for a in range(local_world):
            for b in range(a + 1, local_world):
                sa, sb = set(plans[a][0]), set(plans[b][0])
                if "slice" in plans[a][1] and "slice" in plans[b][1] and (nodes[a] == nodes[b]) and sa != sb:
                    assert sa.isdisjoint(sb), (a, b, plans[a], plans[b])
    check()
