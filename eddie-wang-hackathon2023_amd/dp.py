"""Data-parallel sharding of utterances over the GPUs of one node (one process per GPU).

The reference is batch-1, world-size-1 (W/run.py:38-41, W/build.py:159,222,234): it has no
multi-GPU Whisper path.  Each 30 s clip is an independent unit (encoder, cross K/V and the decode
loop share nothing between clips), so the path shards by utterance with NO data-path collective:
a full weight replica per GPU, rank r takes a contiguous slice of the clips, and the only exchanges
are (i) handing each rank its mel slice and (ii) gathering token ids / log-probs at the end.  With
torch.distributed the backend "nccl" is RCCL over xGMI on this platform; "gloo" is used by the CPU
tests.  Nothing is reduced, so there is no ring all-reduce anywhere.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of `n_items` for `rank`; the first n % world ranks get one more."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def scatter_utterances(mels: Optional[torch.Tensor], n_items: int, feature_shape: Tuple[int, ...],
                       dtype: torch.dtype, device) -> torch.Tensor:
    """Rank 0 holds `mels` [n_items, *feature_shape]; every rank returns its own slice on `device`.
    Ragged slices are padded to the widest one for the collective and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        return mels.to(device)
    rank, world = dist.get_rank(), dist.get_world_size()
    widest = max(shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world))
    mine = torch.empty((widest, *feature_shape), dtype=dtype, device=device)
    chunks = None
    if rank == 0:
        chunks = []
        for r in range(world):
            lo, hi = shard_bounds(n_items, r, world)
            c = torch.zeros((widest, *feature_shape), dtype=dtype, device=device)
            c[: hi - lo] = mels[lo:hi].to(device)
            chunks.append(c)
    dist.scatter(mine, chunks, src=0)
    lo, hi = shard_bounds(n_items, rank, world)
    return mine[: hi - lo]


def gather_results(tokens: torch.Tensor, sum_logprobs: torch.Tensor, n_items: int, width: int, pad_value: int
                   ) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """All ranks send their [n_local, <=width] token rows and [n_local] log-probs; rank 0 returns
    ([n_items, width] int64, [n_items] fp32) in utterance order, other ranks None."""
    if not (dist.is_available() and dist.is_initialized()):
        out = torch.full((tokens.shape[0], width), pad_value, dtype=torch.int64, device=tokens.device)
        out[:, : tokens.shape[1]] = tokens
        return out, sum_logprobs
    rank, world = dist.get_rank(), dist.get_world_size()
    widest = max(shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world))
    tok = torch.full((widest, width), pad_value, dtype=torch.int64, device=tokens.device)
    tok[: tokens.shape[0], : tokens.shape[1]] = tokens
    lp = torch.zeros(widest, dtype=torch.float32, device=tokens.device)
    lp[: sum_logprobs.shape[0]] = sum_logprobs
    tok_list = [torch.empty_like(tok) for _ in range(world)] if rank == 0 else None
    lp_list = [torch.empty_like(lp) for _ in range(world)] if rank == 0 else None
    dist.gather(tok, tok_list, dst=0)
    dist.gather(lp, lp_list, dst=0)
    if rank != 0:
        return None
    toks, lps = [], []
    for r in range(world):
        lo, hi = shard_bounds(n_items, r, world)
        toks.append(tok_list[r][: hi - lo])
        lps.append(lp_list[r][: hi - lo])
    return torch.cat(toks), torch.cat(lps)


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def all_ranks(value: float, device) -> List[float]:
    """`value` of every rank, in rank order, on every rank (one small all-gather; [value] without a process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x[0]) for x in out]


# ---- host placement of a rank (with nothing reduced on the data path, what limits the scaling of utterance DP is
# host-side scheduling -- eight Python processes, each issuing graph replays and a helper thread's encoder layers) ---------------------
def _parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _visible_index(local_rank: int) -> int:
    """Index of this rank's GPU among ALL of the node's GPUs: through HIP_/ROCR_/CUDA_VISIBLE_DEVICES when a launcher has set one."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                ids = [int(x) for x in v.split(",") if x.strip() != ""]
                return ids[local_rank] if local_rank < len(ids) else local_rank
            except ValueError:
                return local_rank              # UUIDs: no way to map without a GPU call -- identity
    return local_rank


def gpu_numa_node(gpu_index: int, sysfs: str = "/sys") -> Optional[int]:
    """NUMA node of the node's `gpu_index`-th GPU, from the KFD topology (GPU nodes = entries with simd_count > 0, in node order = the
    HIP device order) and the DRM device's `numa_node`.  No GPU call: this runs BEFORE the process initialises the runtime.  None when
    sysfs does not say (containers, single-socket boxes that report -1)."""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    try:
        minors = []
        for n in sorted((d for d in os.listdir(base) if d.isdigit()), key=int):
            props = {}
            with open(os.path.join(base, n, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
            if int(props.get("simd_count", "0")) > 0:
                minors.append(int(props.get("drm_render_minor", "-1")))
        if gpu_index >= len(minors) or minors[gpu_index] < 0:
            return None
        with open(os.path.join(sysfs, f"class/drm/renderD{minors[gpu_index]}/device/numa_node")) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError):
        return None


MIN_CPUS_PER_RANK = 4      # a rank = the Python thread that replays the graphs, the encoder's helper thread, RCCL's proxy threads: never squeeze it below this


def plan_rank_cpus(local_rank: int, local_world: int, allowed: Sequence[int], nodes: Sequence[Optional[int]],
                   node_cpus: Dict[int, List[int]]) -> Tuple[List[int], str]:
    """The CPUs rank `local_rank` of `local_world` ranks on this host is pinned to (pure planning: testable without sysfs).
    `allowed` = the CPUs the process may use now, `nodes[r]` = NUMA node of rank r's GPU (None: unknown), `node_cpus` = node -> CPUs.
    A rank whose GPU's node is known gets that node's allowed CPUs, cut into disjoint contiguous slices among the ranks that share the
    node (8 GPUs on 2 sockets: 4 ranks per node, a quarter of the node's cores each; never fewer than MIN_CPUS_PER_RANK CPUs -- then the whole node).
    Unknown node: an even contiguous slice of everything allowed (ranks stay apart, locality is left to the OS) -- unless that
    would leave a rank fewer than MIN_CPUS_PER_RANK CPUs: then nothing is pinned."""
    allowed_sorted = sorted(allowed)
    node = nodes[local_rank] if local_rank < len(nodes) else None
    if node is not None and node in node_cpus:
        mine = [c for c in node_cpus[node] if c in set(allowed_sorted)]
        sharers = [r for r in range(local_world) if r < len(nodes) and nodes[r] == node]
        if mine and local_rank in sharers:
            k, n = sharers.index(local_rank), len(sharers)
            lo, hi = k * len(mine) // n, (k + 1) * len(mine) // n
            if hi - lo >= MIN_CPUS_PER_RANK:
                return mine[lo:hi], f"NUMA node {node} of the rank's GPU, slice {k + 1}/{n} of its {len(mine)} CPUs"
            return mine, f"NUMA node {node} of the rank's GPU (all {len(mine)} CPUs: too few to cut {n} ways)"
    lo, hi = local_rank * len(allowed_sorted) // local_world, (local_rank + 1) * len(allowed_sorted) // local_world
    if hi - lo >= MIN_CPUS_PER_RANK:
        return allowed_sorted[lo:hi], f"GPU NUMA node unknown: even slice {local_rank + 1}/{local_world} of the {len(allowed_sorted)} allowed CPUs"
    return allowed_sorted, (f"GPU NUMA node unknown and fewer than {MIN_CPUS_PER_RANK} CPUs per rank ({len(allowed_sorted)} for {local_world}): not pinned "
                            "(a rank is a Python thread, the encoder's helper thread and RCCL's proxy threads)")


def pin_rank_to_gpu_numa(local_rank: int, local_world: int, sysfs: str = "/sys") -> dict:
    """os.sched_setaffinity of THIS process to the CPUs next to its GPU -- call it before the first GPU call (threads the runtime and
    torch start later inherit the mask; `numactl` / `taskset` in front of a GPU process is a re-exec this pool forbids).  Returns what
    was done, for the bench line.  Never raises: placement is a nicety, not part of the compute path."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        nodes = [gpu_numa_node(_visible_index(r), sysfs) for r in range(local_world)]
        node_cpus: Dict[int, List[int]] = {}
        for n in {x for x in nodes if x is not None}:
            try:
                with open(os.path.join(sysfs, f"devices/system/node/node{n}/cpulist")) as f:
                    node_cpus[n] = _parse_cpulist(f.read())
            except (OSError, ValueError):
                pass
        cpus, why = plan_rank_cpus(local_rank, local_world, allowed, nodes, node_cpus)
        if cpus and set(cpus) != set(allowed):
            os.sched_setaffinity(0, cpus)
        return {"pinned": bool(cpus) and set(cpus) != set(allowed), "cpus": len(cpus), "first_cpu": cpus[0] if cpus else None,
                "last_cpu": cpus[-1] if cpus else None, "gpu_numa_node": nodes[local_rank] if local_rank < len(nodes) else None, "how": why}
    except (OSError, AttributeError, ValueError) as e:      # (no sched_setaffinity on this platform, a cpuset that forbids it, ...)
        return {"pinned": False, "cpus": None, "how": f"not pinned: {e}"}


def rank_seed(seed: int, rank: int) -> int:
    """Seed of rank `rank`'s own shard of a synthetic batch: (seed, rank) -> one integer; rank 0 keeps `seed` itself, so a one-rank job
    draws exactly what it always drew."""
    return int(seed) + 1_000_003 * int(rank)
