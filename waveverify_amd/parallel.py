"""Data-parallel sharding of clip batches over the GPUs of one node.

Clips are independent units in all three nets (no batch statistics, FiLM is per clip — SURVEY.md
section 8e), so the batch dimension is split contiguously, weights are replicated (56 MB) and the
forward passes need NO collective.  One process per GPU (torch.distributed, backend "nccl" = RCCL
on ROCm, "gloo" in CPU tests); the only communication is the optional gather of the [B,16] bit
decisions / mean probabilities onto every rank.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first n % world ranks get one extra clip."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather per-rank row blocks (possibly ragged by one row) into the full [n_total, ...]."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rows = max(shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0]
               for r in range(world))
    pad = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        parts.append(out[r][: hi - lo])
    return torch.cat(parts, dim=0)


def embed_detect_sharded(embed: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                         detect: Callable[[torch.Tensor], torch.Tensor],
                         audio: torch.Tensor, message: torch.Tensor, rank: int, world: int,
                         gather: bool = True, group=None
                         ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Run embed -> detect on this rank's shard of (audio [B,1,T], message [B,16]).
    Returns (local watermarked audio, local mean-prob [b,16], gathered mean-prob [B,16] or None)."""
    x, m = shard(audio, rank, world), shard(message, rank, world)
    wm = embed(x, m)
    mp = detect(wm)
    return wm, mp, (all_gather_rows(mp, audio.shape[0], group) if gather else None)
