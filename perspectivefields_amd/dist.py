"""Image-level data parallelism: one process per GPU, images are independent units
(no cross-image op on the path; BatchNorm is eval-mode), weights replicated.  The only
exchange step is an all-gather of the per-image ParamNet scalars ((B_local, 8) fp32,
<= 8 KB at B=256) over RCCL/xGMI (`torch.distributed` backend "nccl"), or gloo on CPU in
tests.  Dense fields stay on the GPU that produced them, as in the reference (outputs
stay on device, perspectivefields.py:255-272)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_round_robin_by_bucket(sizes: Sequence[Tuple[int, int]], rank: int, world: int) -> List[int]:
    """Mixed-resolution streams: round-robin within each (H, W) bucket so every rank gets the same
    mix of post-process work (the network batch itself is resolution independent: all -> 320x320)."""
    buckets = {}
    for i, hw in enumerate(sizes):
        buckets.setdefault(tuple(hw), []).append(i)
    mine, k = [], 0
    for hw in sorted(buckets):
        for i in buckets[hw]:
            if k % world == rank:
                mine.append(i)
            k += 1
    return sorted(mine)


def gather_params(local: torch.Tensor, counts: Sequence[int] = None) -> torch.Tensor:
    """All-gather (B_local, P) -> (sum B_local, P) in rank order.  `counts` = per-rank row counts
    (needed only when shards are ragged); single-process runs return `local` unchanged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if counts is None:
        counts = [local.shape[0]] * world
    mx = max(counts)
    padded = local
    if local.shape[0] < mx:
        padded = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    chunks = [out[r * mx : r * mx + counts[r]] for r in range(world)]
    return torch.cat(chunks)
