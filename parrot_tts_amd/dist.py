"""One process per GPU; utterances shard by batch row; the only collective is the final gather.

Works with any torch.distributed backend: "nccl" (= RCCL over xGMI on MI355X) in production,
"gloo" in the CPU tests."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  A single process (no WORLD_SIZE) needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_rows(n_rows: int, rank: int, world: int) -> slice:
    """Contiguous row range of rank `rank` (first n_rows % world ranks get one extra row)."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Rows [start, stop) of every (B, ...) tensor in a collated batch.  NB the TTE's results depend on
    the PADDED batch shape (reference quirks Q1/Q7), so shard pre-padded buckets, never re-pad."""
    n = next(v.shape[0] for v in batch.values() if isinstance(v, torch.Tensor))
    sl = shard_rows(n, rank, world)
    return {k: (v[sl] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}


def gather_waveforms(wav: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather equal-shaped (b, 1, n) waveform shards to rank `dst` in rank order -> (world*b, 1, n) there,
    None elsewhere.  One RCCL gather: every peer has its own xGMI link to the root, 21 MB per shard."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return wav
    world, rank = dist.get_world_size(), dist.get_rank()
    out: Optional[List[torch.Tensor]] = None
    if rank == dst:
        out = [torch.empty_like(wav) for _ in range(world)]
    dist.gather(wav.contiguous(), out, dst=dst)
    return torch.cat(out, dim=0) if rank == dst else None


def gather_ragged_rows(rows: List[List[int]], dst: int = 0) -> Optional[List[List[int]]]:
    """Gather ragged python lists (unit ids) to `dst` in rank order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(rows, out, dst=dst)
    return [r for part in out for r in part] if out is not None else None
