"""Data-parallel helpers (one process per GPU; torch.distributed backend 'nccl' == RCCL over xGMI on ROCm).

The ELBO path shards by field pair: every loss term is a mean over the batch and GroupNorm is per sample
(SURVEY.md §8e), so the global-batch gradient is the mean of the equal-sized shard gradients.  The only exchange is
one all-reduce of the flat fp32 gradient buffer per step.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Initialise the default process group from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* variables."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def allreduce_mean_(flat: torch.Tensor, group=None, bucket_elems: int = 0, average: bool = True,
                    wire_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """In-place mean over the process group of a flat gradient buffer.  bucket_elems > 0 splits the reduction into
    buckets of that many elements (point-to-point xGMI rings are per-link bound, so a few large buckets are best);
    0 = one collective.  average=False leaves the SUM in place (the caller folds 1 / world into a pass it makes anyway).
    wire_dtype (e.g. torch.bfloat16): the optional gradient compression of SURVEY.md §8e - the buffer is cast to that type for
    the collective and the reduced values are written back in fp32 (half the bytes on the wire, ~3 significant digits)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return flat
    wire = flat if wire_dtype is None or wire_dtype == flat.dtype else flat.to(wire_dtype)
    if bucket_elems <= 0 or bucket_elems >= wire.numel():
        dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=group)
    else:
        works = [dist.all_reduce(wire[i:i + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True)
                 for i in range(0, wire.numel(), bucket_elems)]
        for w in works:
            w.wait()
    if wire is not flat:
        flat.copy_(wire)
    if average:
        flat.mul_(1.0 / world)
    return flat


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous shard of a global batch (global batch 256 -> 32 field pairs per rank at world 8)."""
    n = x.shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by the world size {world}")
    per = n // world
    return x[rank * per:(rank + 1) * per]


def rank_seed(base_seed: int, rank: int) -> int:
    """Rank-offset seed for the reparameterisation noise and dropout masks (no communication)."""
    return (int(base_seed) + 0x9E3779B9 * (rank + 1)) & 0x7FFFFFFFFFFFFFFF
