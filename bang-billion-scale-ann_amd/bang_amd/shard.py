"""Query sharding across the GPUs of one node (no counterpart in the single-GPU reference).

Queries are independent, so rank r of W searches the contiguous range [r*Q/W, (r+1)*Q/W) on its own replica
of the index and ONE collective -- an all-gather of the [Q/W][k] id blocks (RCCL over xGMI on GPUs, gloo in
the CPU tests) -- rebuilds the [Q][k] result on every rank.  ~100 KB per rank at Q=10K, k=10: latency bound."""
from __future__ import annotations

import numpy as np


def shard_range(Q: int, rank: int, world: int):
    return Q * rank // world, Q * (rank + 1) // world


def gather_ids(ids_local: np.ndarray, Q: int, k: int, rank: int, world: int, device=None):
    """All-gather the per-rank id blocks into the full [Q][k] u64 array.  Uses torch.distributed's default
    process group (backend "nccl" == RCCL on ROCm, or "gloo")."""
    if world == 1:
        return ids_local
    import torch
    import torch.distributed as dist
    pad = (Q + world - 1) // world
    dev = device if device is not None else torch.device("cpu")
    mine = torch.zeros((pad, k), dtype=torch.int64, device=dev)
    mine[: ids_local.shape[0]] = torch.from_numpy(np.ascontiguousarray(ids_local).view(np.int64)).to(dev)
    allv = torch.empty((world * pad, k), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allv, mine)
    allv = allv.cpu().numpy().view(np.uint64)
    out = np.empty((Q, k), dtype=np.uint64)
    for r in range(world):
        a, b = shard_range(Q, r, world)
        out[a:b] = allv[r * pad: r * pad + (b - a)]
    return out
