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


# ---- training-step gradient all-reduce (BASELINE configs[2]; reference: DDP in scripts/train.py:875-876,1277,1347)
# The forward passes need no collective; the only exchange step of the reference's data-parallel training
# is the per-step all-reduce of the fp32 gradients: 56.1 MB (generator + detector + locator) and 170.1 MB
# (discriminator), SURVEY.md section 2.2.  xGMI is point-to-point (one ~153 GB/s link per peer), so a ring
# all-reduce is per-link bound: buckets of >= 25 MB keep the link busy and let bucket i's reduction overlap
# the backward of the layers behind it.
GRAD_PAYLOAD_BYTES = {"generator+detector+locator": 56_100_000, "discriminator": 170_100_000}
DEFAULT_BUCKET_BYTES = 25 * 1024 * 1024


def plan_buckets(numels, bucket_bytes: int = DEFAULT_BUCKET_BYTES, itemsize: int = 4):
    """Greedy DDP-style bucketing of a parameter list (in reverse order: gradients become ready from the
    last layer backwards).  -> list of lists of parameter indices; a bucket closes once it holds at least
    `bucket_bytes`; a single tensor larger than that is a bucket of its own."""
    if bucket_bytes < 1:
        raise ValueError("bucket_bytes must be positive")
    buckets, cur, cur_bytes = [], [], 0
    for i in reversed(range(len(numels))):
        cur.append(i)
        cur_bytes += int(numels[i]) * itemsize
        if cur_bytes >= bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
    if cur:
        buckets.append(cur)
    return buckets


def allreduce_mean_(grads, buckets, group=None, async_op: bool = True):
    """In-place mean all-reduce of `grads` (list of tensors), one flat collective per bucket.  Buckets are
    launched back to back (async) and waited for at the end, so bucket i+1's transfer is queued while
    bucket i is still on the links.  Returns the flat buffers' handles' count (for tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    flats, works = [], []
    for b in buckets:
        flat = torch.cat([grads[i].reshape(-1) for i in b]) if len(b) > 1 else grads[b[0]].reshape(-1).clone()
        flats.append(flat)
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op))
    for w in works:
        if w is not None:
            w.wait()
    for b, flat in zip(buckets, flats):
        flat.div_(world)
        off = 0
        for i in b:
            n = grads[i].numel()
            grads[i].copy_(flat[off:off + n].view_as(grads[i]))
            off += n
    return len(flats)


def allreduce_mean_flat_(arena: torch.Tensor, bucket_bytes: int = DEFAULT_BUCKET_BYTES, group=None) -> int:
    """In-place mean all-reduce of a FLAT gradient arena (waveverify_amd/train.py keeps a net's gradients contiguous):
    buckets are plain slices -- no gather / scatter copies -- launched back to back from the END of the arena
    (backward fills it last layer first) and waited for together.  Returns the number of collectives."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    if arena.dim() != 1 or not arena.is_contiguous():
        raise ValueError("flat contiguous arena required")
    world = dist.get_world_size(group)
    per = max(1, bucket_bytes // arena.element_size())
    works, hi = [], arena.numel()
    while hi > 0:
        lo = max(0, hi - per)
        works.append(dist.all_reduce(arena[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
        hi = lo
    for w in works:
        w.wait()
    arena.div_(world)
    return len(works)


class OverlappedFlatReducer:
    """The same mean all-reduce as `allreduce_mean_flat_` -- identical buckets (slices cut from the END of the arena), identical
    collectives, identical result -- but every bucket is launched as soon as backward has written the last gradient inside it, so
    that the transfer runs under the rest of the backward pass (DDP's overlap, /root/reference/scripts/train.py:875-876,1277,1347).

        red = OverlappedFlatReducer(grads, ranges)     # ranges: {parameter key: (lo, hi)} offsets into the arena
        ... backward ...; red.mark(keys just written) after every stage ...
        red.flush()                                    # launch whatever is still pending (nothing, when every key was marked)
        red.wait()                                     # before the optimizer reads the arena: waits, then divides by the world size

    Every rank marks in the same order (same code path), so the collectives are issued in the same order everywhere.  With one
    rank (or no process group) nothing is launched and the arena is left as it is."""

    def __init__(self, arena: torch.Tensor, ranges, bucket_bytes: int = DEFAULT_BUCKET_BYTES, group=None):
        import torch.distributed as dist
        if arena.dim() != 1 or not arena.is_contiguous():
            raise ValueError("flat contiguous arena required")
        self.arena, self.group = arena, group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.active else 1
        per = max(1, bucket_bytes // arena.element_size())
        self.buckets, hi = [], arena.numel()                      # [(lo, hi)], last layers first
        while hi > 0:
            lo = max(0, hi - per)
            self.buckets.append((lo, hi))
            hi = lo
        self.pending = [set() for _ in self.buckets]              # keys not yet written, per bucket
        self.where = {}                                           # key -> buckets it touches
        for k, (lo, hi) in ranges.items():
            if hi <= lo:
                continue
            hit = [i for i, (blo, bhi) in enumerate(self.buckets) if lo < bhi and hi > blo]
            self.where[k] = hit
            for i in hit:
                self.pending[i].add(k)
        self.launched = [False] * len(self.buckets)
        self.works = []

    def _launch(self, i: int) -> None:
        import torch.distributed as dist
        lo, hi = self.buckets[i]
        self.launched[i] = True
        if self.active:
            self.works.append(dist.all_reduce(self.arena[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def mark(self, keys) -> int:
        """The gradients of `keys` are final.  Launches every bucket this completes; returns how many."""
        touched = set()
        for k in keys:
            for i in self.where.get(k, ()):
                self.pending[i].discard(k)
                touched.add(i)
        n = 0
        for i in sorted(touched):                                 # ascending bucket index = from the end of the arena
            if not self.launched[i] and not self.pending[i]:
                self._launch(i)
                n += 1
        return n

    def flush(self) -> int:
        n = 0
        for i in range(len(self.buckets)):
            if not self.launched[i]:
                self._launch(i)
                n += 1
        return n

    def wait(self) -> int:
        """Complete the reduction (flushes first).  Returns the number of collectives that ran."""
        self.flush()
        for w in self.works:
            w.wait()
        n = len(self.works)
        self.works = []
        if self.active:
            self.arena.div_(self.world)
        return n
