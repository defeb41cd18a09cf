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
        backend = os.environ.get("PARROT_DIST_BACKEND") or backend
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def local_device(local_rank: int) -> torch.device:
    """cuda:<local_rank>, wrapped onto the devices this process can see: with fewer GPUs than ranks (the 2-ranks-on-one-GPU
    test of the N>1 path; RCCL refuses two ranks on one device, so that test sets PARROT_DIST_BACKEND=gloo) ranks share."""
    n = torch.cuda.device_count()
    return torch.device("cuda", local_rank % n if n else local_rank)


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


_meta_group = None          # gloo side group for host-side shape exchange (no device sync)
_gather_buf: Dict[tuple, torch.Tensor] = {}


def _meta_pg():
    """A CPU (gloo) process group beside the RCCL one: shapes are host integers, and exchanging them through the device
    backend would cost a stream synchronisation per step."""
    global _meta_group
    if dist.get_backend() == "gloo":
        return None  # the default group already is one
    if _meta_group is None:
        _meta_group = dist.new_group(backend="gloo")
    return _meta_group


def gather_waveforms(wav: torch.Tensor, dst: int = 0, equal_shapes: bool = False,
                     n_samples: Optional[torch.Tensor] = None):
    """Gather (b_r, 1, n_r) waveform shards to rank ``dst`` in rank order -> (sum b_r, 1, max n_r) there, None elsewhere.
    One RCCL gather (grouped send/recv: every peer has its own xGMI link to the root; 21 MB per shard at B=64 x 256
    units) into ONE receive buffer that is allocated once per shape and reused -- the returned tensor is a view of it and
    is overwritten by the next call with the same shapes; the counterpart of the result queue of reference
    utils/vocoder/inference.py:201-205,255.

    Shards may differ in rows (B % world != 0) and in length (L = max over the shard's rows is data dependent): the
    (b_r, n_r) pairs -- host integers -- are exchanged first over a gloo side group, shorter shards are zero-padded to
    the longest.  ``equal_shapes=True`` skips that exchange when the caller knows all shards agree (fixed-length
    workloads).  With ``n_samples`` ((b_r,) valid samples per row) the root gets ``(wav, n_samples_all)``."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return wav if n_samples is None else (wav, n_samples)
    world, rank = dist.get_world_size(), dist.get_rank()
    b, n = int(wav.shape[0]), int(wav.shape[-1])
    if equal_shapes:
        shapes = [(b, n)] * world
    else:
        mine = torch.tensor([b, n], dtype=torch.int64)
        allsh = [torch.empty(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allsh, mine, group=_meta_pg())
        shapes = [(int(t[0]), int(t[1])) for t in allsh]
    bmax, nmax = max(s[0] for s in shapes), max(s[1] for s in shapes)
    host_staged = dist.get_backend() == "gloo" and wav.device.type != "cpu"  # gloo's gather takes CPU tensors only
    send = wav.reshape(b, n)
    if (b, n) != (bmax, nmax):
        send = torch.zeros((bmax, nmax), dtype=wav.dtype, device=wav.device)
        send[:b, :n] = wav.reshape(b, n)
    send = send.contiguous()
    if host_staged:
        send = send.cpu()
    out = None
    if rank == dst:
        key = (world, bmax, nmax, wav.dtype, str(send.device))
        if key not in _gather_buf:
            _gather_buf.clear()  # one live shape: do not pin the memory of shapes long gone
            _gather_buf[key] = torch.empty((world, bmax, nmax), dtype=wav.dtype, device=send.device)
        out = list(_gather_buf[key].unbind(0))
    _collect(send, out, dst, key if rank == dst else None)
    lens_all = None
    if n_samples is not None:
        lens_all = [torch.empty(s[0], dtype=torch.int64) for s in shapes] if rank == dst else None
        lens_all = _gather_host_rows(n_samples.to("cpu", torch.int64), lens_all, shapes, dst)
    if rank != dst:
        return None
    buf = _gather_buf[key]
    if all(s[0] == bmax for s in shapes):
        res = buf.view(world * bmax, 1, nmax)          # no copy: rows already in rank order
    else:
        res = torch.cat([buf[r, : shapes[r][0]] for r in range(world)], dim=0).unsqueeze(1)
    if host_staged:
        res = res.to(wav.device)
    return res if n_samples is None else (res, lens_all)


_gather_mode = os.environ.get("PARROT_GATHER", "auto")  # auto | gather | allgather
_gather_latched = False  # "gather" was reached by the ranks' agreement in auto mode (not forced through the environment)
GATHER_FATAL_EXIT_CODE = 70  # exit status of a rank whose latched gather failed (see _collect)
_allgather_buf: Dict[tuple, torch.Tensor] = {}


def _collect(send: torch.Tensor, out, dst: int, key) -> None:
    """Move every rank's (bmax, nmax) shard into the root's receive buffer.  `dist.gather` into the `unbind` views of ONE
    contiguous buffer (grouped send / recv under RCCL: each peer uses its own xGMI link to the root).  Should a backend
    reject the view list, PARROT_GATHER=auto falls back -- once, for the rest of the process, on all ranks together -- to
    `all_gather_into_tensor` on a single contiguous tensor (every rank then holds a copy: world x 21 MB at B = 64)."""
    global _gather_mode, _gather_latched
    world, rank = dist.get_world_size(), dist.get_rank()
    if _gather_mode == "gather":
        # Latched (or forced) mode: no per-step agreement any more, so an error on ONE rank (a new shape, OOM, a backend hiccup)
        # would leave its peers blocked inside the collective until the backend's watchdog fires.  It is fatal for the whole
        # group by design: the failing rank reports and exits (PARROT_GATHER_FATAL=raise re-raises instead, for callers that
        # tear the group down themselves) -- under torchrun / mp.spawn the launcher then ends the peers at once.
        try:
            dist.gather(send, out, dst=dst)
        except RuntimeError as e:
            # default: end the process only under a launcher that tears the group down (torchrun sets TORCHELASTIC_RUN_ID) and only
            # when the mode was LATCHED by an earlier agreement; a user-forced PARROT_GATHER=gather, a notebook or a bare
            # single-node session gets the exception (atexit handlers, finally blocks and buffered output intact)
            fatal = os.environ.get("PARROT_GATHER_FATAL") or ("exit" if (_gather_latched and os.environ.get("TORCHELASTIC_RUN_ID")) else "raise")
            if fatal == "raise":
                raise
            import sys
            sys.stdout.flush()
            print(f"parrot_tts_amd.dist: rank {rank}: dist.gather failed after the gather mode was latched ({e}); the peers are "
                  "inside the collective -- ending this process so that the launcher tears the group down", file=sys.stderr, flush=True)
            os._exit(GATHER_FATAL_EXIT_CODE)
        return
    if _gather_mode == "auto":
        ok = 1
        try:
            dist.gather(send, out, dst=dst)
        except RuntimeError:
            ok = 0
        # all ranks must agree on the fallback (a failure on the root only would otherwise deadlock the next call).  The
        # agreement is reached ONCE: after the first call every rank latches the mode, so later steps carry no host-side
        # all_reduce (a cross-rank host barrier per step would stop the host from running ahead of the device)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=_meta_pg())
        if int(flag) == 1:
            _gather_mode, _gather_latched = "gather", True
            return
        _gather_mode = "allgather"
        if rank == 0:
            print("parrot_tts_amd.dist: dist.gather into buffer views failed; using all_gather_into_tensor from now on", flush=True)
    k2 = (world,) + tuple(send.shape) + (send.dtype, str(send.device))
    if k2 not in _allgather_buf:
        _allgather_buf.clear()
        _allgather_buf[k2] = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
    full = _allgather_buf[k2]
    dist.all_gather_into_tensor(full, send)
    if rank == dst:
        _gather_buf[key].copy_(full)


def dist_info() -> dict:
    """What the process group actually is (for bench.py's JSON line): backend, ranks, devices visible to this process."""
    if not dist.is_initialized():
        return {"backend": None, "nranks": 1, "device_count": torch.cuda.device_count()}
    return {"backend": dist.get_backend(), "nranks": dist.get_world_size(), "device_count": torch.cuda.device_count(),
            "gather": _gather_mode}


def _gather_host_rows(mine: torch.Tensor, out, shapes, dst):
    """Ragged CPU int64 rows to ``dst`` over the gloo side group (padded to the longest shard)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    bmax = max(s[0] for s in shapes)
    pad = torch.zeros(bmax, dtype=torch.int64)
    pad[: mine.numel()] = mine
    recv = [torch.empty(bmax, dtype=torch.int64) for _ in range(world)] if rank == dst else None
    dist.gather(pad, recv, dst=dst, group=_meta_pg())
    if rank != dst:
        return None
    return torch.cat([recv[r][: shapes[r][0]] for r in range(world)])


def gather_ragged_rows(rows: List[List[int]], dst: int = 0) -> Optional[List[List[int]]]:
    """Gather ragged python lists (unit ids) to `dst` in rank order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(rows, out, dst=dst)
    return [r for part in out for r in part] if out is not None else None
