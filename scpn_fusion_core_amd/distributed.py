"""Ensemble sharding across the GPUs of one node: one process per GPU, no data-path collective.

The equilibria of an ensemble (coil / shape scan, shot replay) are independent units, so the path shards across
members: rank r owns a contiguous block of ``ceil(total / world)`` members, solves it with its resident
:class:`~scpn_fusion_core_amd.ensemble.EnsembleSolver` and the only exchange step is the final gather of the psi
blocks and the per-member scalar table (``torch.distributed`` all_gather: RCCL over xGMI with backend "nccl", gloo
in the CPU tests).  torch is imported lazily -- it is plumbing for the rendezvous / collective only.
"""

from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [start, stop) of ``total`` members owned by ``rank`` (blocks of ceil(total/world))."""
    if total < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("need total >= 0, world >= 1 and 0 <= rank < world")
    per = -(-total // world)
    start = min(rank * per, total)
    return start, min(start + per, total)


def shard_sizes(total: int, world: int) -> list[int]:
    return [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]


class DevicePointer:
    """Wraps a raw device address as a ``__cuda_array_interface__`` object so torch can alias it without a copy."""

    def __init__(self, ptr: int, shape: tuple[int, ...], typestr: str = "<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3,
                                         "strides": None}


def gather_blocks(local, total: int, group=None):
    """All-gather equally-padded member blocks; returns the (total, ...) array on every rank.

    ``local`` is a torch tensor (CPU for gloo, device for nccl) or a numpy array of shape (n_local, ...).
    Blocks are padded to ceil(total/world) members so a single all_gather_into_tensor suffices.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    was_numpy = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if was_numpy else local
    per = -(-total // world)
    start, stop = shard_range(total, rank, world)
    if t.shape[0] != stop - start:
        raise ValueError(f"rank {rank} holds {t.shape[0]} members, expected {stop - start}")
    if t.shape[0] < per:  # pad the last rank(s)
        pad = torch.zeros((per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    out = torch.empty((world * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    out = out[:total]
    return out.numpy() if was_numpy else out


def gather_blocks_async(local, total: int, out=None, group=None):
    """Non-blocking form of :func:`gather_blocks` for torch tensors: launches the all-gather and returns ``(work, out)``
    immediately so the collective (RCCL over xGMI) overlaps the next solve.  ``work.wait()`` followed by a synchronisation of
    the current torch stream must precede any reuse of ``local``'s storage or of ``out`` (pass the previous ``out`` back in
    to reuse its allocation).  Members are padded per rank exactly as in :func:`gather_blocks`; ``out[:total]`` is the result."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-total // world)
    start, stop = shard_range(total, rank, world)
    t = local
    if t.shape[0] != stop - start:
        raise ValueError(f"rank {rank} holds {t.shape[0]} members, expected {stop - start}")
    if t.shape[0] < per:
        pad = torch.zeros((per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    shape = (world * per,) + tuple(t.shape[1:])
    if out is None or tuple(out.shape) != shape or out.dtype != t.dtype or out.device != t.device:
        out = torch.empty(shape, dtype=t.dtype, device=t.device)
    work = dist.all_gather_into_tensor(out, t.contiguous(), group=group, async_op=True)
    return work, out
