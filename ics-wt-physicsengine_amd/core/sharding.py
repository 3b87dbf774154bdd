"""Instance-parallel sharding of a reactor ensemble over the GPUs of one node.

Reactors are independent ODE systems, so the ensemble
is cut into contiguous blocks of reactor indices, one per rank, and stepping
needs no communication.  The only collective is the final state gather
(``all_gather`` = RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU
tests).
"""
from __future__ import annotations

from typing import Tuple


def shard_bounds(n_reactors: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of reactor indices owned by ``rank``; the first
    ``n_reactors % world_size`` ranks get one extra reactor."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(n_reactors, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_state(local, world_size: int, force_collective: bool = False, sizes=None):
    """All-gather per-rank state tensors (3, N_local, n) into reactor order (3, N_total, n).
    ``local`` is a torch tensor on the backend's device.  ``sizes`` = N_local of every rank
    (what ``shard_bounds`` hands out; unequal when N_total % world_size != 0): shards are padded to
    the largest block for the collective and trimmed afterwards.  Without ``sizes`` every rank must
    hold the same number of reactors."""
    import torch
    import torch.distributed as dist

    if world_size == 1 and not force_collective:
        return local
    if sizes is None:
        sizes = [int(local.shape[1])] * world_size
    sizes = [int(s) for s in sizes]
    if len(sizes) != world_size:
        raise ValueError("sizes must have one entry per rank")
    pad = max(sizes)
    if local.shape[1] > pad:
        raise ValueError("local shard is larger than any entry of sizes")
    if local.shape[1] < pad:     # pad to ceil(N / W) reactors
        padded = torch.zeros((local.shape[0], pad, local.shape[2]), dtype=local.dtype, device=local.device)
        padded[:, :local.shape[1]] = local
    else:
        padded = local
    flat = padded.contiguous().view(-1)
    out = torch.empty(world_size * flat.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, flat)
    out = out.view(world_size, local.shape[0], pad, local.shape[2])
    if all(s == pad for s in sizes):
        # (W, 3, Nl, n) -> (3, W*Nl, n)
        return out.permute(1, 0, 2, 3).reshape(local.shape[0], world_size * pad, local.shape[2])
    return torch.cat([out[r, :, :sizes[r]] for r in range(world_size)], dim=1)
