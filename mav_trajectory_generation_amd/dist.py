"""Multi-GPU sharding of a batch of independent trajectories.

The solve needs no communication: rank r of G owns the contiguous slice [r*B/G, (r+1)*B/G) and runs the same
single-GPU path on it.  The only collective is the optional final gather of the coefficient buffer
(torch.distributed all_gather: RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_range(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split: sizes differ by at most one, concatenation over ranks = range(batch)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_coeffs(local, batch: int, group=None):
    """all_gather the per-rank coefficient slices [b_local][K][D][N] into the full [batch][K][D][N] tensor
    (ranks may hold different slice lengths: padded to the longest, then trimmed)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]
    longest = max(sizes)
    pad = local
    if local.shape[0] < longest:
        pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous(), group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


def solve_sharded(plan, times, d_fixed, layout: str = "aos", gather: bool = False, group=None):
    """Each rank passes the FULL host-side problem description and solves only its shard on its GPU.
    times / d_fixed: full-batch CUDA tensors in `layout`; returns this rank's coeffs (or all, if gather)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    batch = times.shape[0] if layout == "aos" else times.shape[1]
    lo, hi = shard_range(batch, rank, world)
    if layout == "aos":
        t, f = times[lo:hi].contiguous(), d_fixed[lo:hi].contiguous()
    else:
        t, f = times[:, lo:hi].contiguous(), d_fixed[:, :, lo:hi].contiguous()
    coeffs, _, _ = plan.solve(t, f, layout=layout)
    return gather_coeffs(coeffs, batch, group) if gather else coeffs
