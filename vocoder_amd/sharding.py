"""Utterance-batch sharding across the GPUs of one node.

The generator path has no exchange step: clips are independent.  One process per GPU (``torch.distributed``, backend
"nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests) and collectives only at the two ends:

* ``broadcast_state_dict`` — one-time weight fan-out from rank 0 (56 MB fp32 for HiFiGAN-V1);
* ``scatter_batch`` / ``gather_batch`` — optional mel distribution and waveform collection for a caller that holds the
  whole batch on rank 0.  Nothing is exchanged inside the forward, so there is no all-reduce and ring bandwidth never
  binds; a rank that already has its own clips simply calls the generator on its slice (``shard_slice``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_slice(batch: int, world_size: int, rank: int) -> slice:
    """Contiguous, balanced split of ``batch`` clips: the first ``batch % world`` ranks get one extra clip.
    Ragged and empty shards are legal (batch < world_size leaves trailing ranks with nothing)."""
    if world_size < 1 or not 0 <= rank < world_size or batch < 0:
        raise ValueError(f"bad shard request batch={batch} world={world_size} rank={rank}")
    base, extra = divmod(batch, world_size)
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def shard_sizes(batch: int, world_size: int) -> list[int]:
    return [shard_slice(batch, world_size, r).stop - shard_slice(batch, world_size, r).start for r in range(world_size)]


def broadcast_state_dict(state_dict: dict | None, src: int = 0, device=None) -> dict:
    """Rank ``src`` passes its (numpy or tensor) state dict; every rank returns an identical dict of CPU tensors.
    Keys/shapes travel as an object list, payload as ONE flat fp32 buffer (a single large collective, not 291 small
    ones)."""
    rank = dist.get_rank()
    device = torch.device("cpu") if device is None else torch.device(device)
    meta = [None]
    if rank == src:
        items = [(k, torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).float().cpu())
                 for k, v in state_dict.items()]
        meta = [[(k, tuple(t.shape)) for k, t in items]]
    dist.broadcast_object_list(meta, src=src)
    total = sum(int(np.prod(s)) if len(s) else 1 for _, s in meta[0])
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if rank == src:
        flat.copy_(torch.cat([t.reshape(-1) for _, t in items]).to(device))
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    flat = flat.cpu()
    for k, s in meta[0]:
        n = int(np.prod(s)) if len(s) else 1
        out[k] = flat[off:off + n].reshape(s).clone()
        off += n
    return out


def scatter_batch(full: torch.Tensor | None, batch: int, trailing_shape, src: int = 0, device=None) -> torch.Tensor:
    """Rank ``src`` holds ``full`` (batch, *trailing_shape); each rank receives its ``shard_slice``."""
    rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device("cpu") if device is None else torch.device(device)
    sizes = shard_sizes(batch, world)
    mine = torch.empty((sizes[rank],) + tuple(trailing_shape), dtype=torch.float32, device=device)
    reqs = []
    if rank == src:
        for r in range(world):
            part = full[shard_slice(batch, world, r)].contiguous().to(device)
            if r == src:
                mine.copy_(part)
            elif sizes[r]:
                reqs.append(dist.isend(part, dst=r))
    elif sizes[rank]:
        dist.recv(mine, src=src)
    for q in reqs:
        q.wait()
    return mine


def gather_batch(local: torch.Tensor, batch: int, dst: int = 0) -> torch.Tensor | None:
    """Inverse of ``scatter_batch``: rank ``dst`` returns the (batch, ...) concatenation, others None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = shard_sizes(batch, world)
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    if rank == dst:
        parts = []
        for r in range(world):
            if r == dst:
                parts.append(local)
            else:
                buf = torch.empty((sizes[r],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
                if sizes[r]:
                    dist.recv(buf, src=r)
                parts.append(buf)
        return torch.cat(parts, 0)
    if sizes[rank]:
        dist.send(local.contiguous(), dst=dst)
    return None
