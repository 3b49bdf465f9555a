"""Trajectory sharding across the GPUs of one node.

The path shards by independent trajectories: rank r of W owns a contiguous slice of the batch, the
SDF and robot model are replicated per device, there is NO per-iteration exchange, and the only
collective is one all-gather of the results at the end (RCCL on GPUs -- torch.distributed backend
"nccl" -- or gloo for the CPU rehearsal in tests/test_sharding_gloo.py)."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` trajectories for `rank`; the first `total % world`
    ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_problem(arrays: Dict[str, np.ndarray], world: int, rank: int) -> Dict[str, np.ndarray]:
    """Slice every per-trajectory array (leading dimension = batch) for this rank."""
    total = next(iter(arrays.values())).shape[0]
    lo, hi = shard_range(total, world, rank)
    return {k: np.ascontiguousarray(v[lo:hi]) for k, v in arrays.items()}


def gather_results(local, total: int, group=None):
    """All-gather a per-trajectory torch tensor (leading dimension = this rank's slice) into the full
    batch order on every rank.  Slices may be uneven (padded to the largest for the collective)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_range(total, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if dist.get_backend(group) == "nccl" and all(hi - lo == mx for lo, hi in sizes):
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad, group=group)
        return out
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
