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

from typing import List, Optional, Tuple

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
