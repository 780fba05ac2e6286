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


class DeviceGather:
    """The collective of a sharded step with the ids never leaving device memory: the engine writes this rank's [Q/W][k] block into
    `mine` (bang_query_dev_e), ONE all_gather_into_tensor (RCCL over xGMI) fills `all` on every rank, and only the rank that hands
    the batch's answer to its caller copies the gathered block to the host (once).  Buffers are allocated once per allocation, not
    per step.  coll_device != device (gloo dry runs on a shared GPU -- RCCL refuses two ranks on one device): the engine still
    leaves the ids in `mine` on the device; the block is staged to the collective's device for the gather."""

    def __init__(self, Q: int, k: int, rank: int, world: int, device, coll_device=None):
        import torch
        self.Q, self.k, self.rank, self.world = Q, k, rank, world
        self.pad = (Q + world - 1) // world
        self.q0, self.q1 = shard_range(Q, rank, world)
        self.coll_device = coll_device if coll_device is not None else device
        self.mine = torch.zeros((self.pad, k), dtype=torch.int64, device=device)
        self.all = torch.empty((world * self.pad, k), dtype=torch.int64, device=self.coll_device)
        self.dists = torch.zeros((k, self.q1 - self.q0), dtype=torch.float32, device=device)   # this rank's distances [k][Q/W] (not gathered)

    def gather(self):
        import torch.distributed as dist
        src = self.mine if self.mine.device == self.all.device else self.mine.to(self.all.device)
        dist.all_gather_into_tensor(self.all, src)
        return self.all

    def local_ids(self) -> np.ndarray:
        return self.mine[: self.q1 - self.q0].cpu().numpy().view(np.uint64)

    def local_dists(self) -> np.ndarray:
        return self.dists.cpu().numpy()

    def batch_ids(self, copy: bool = True) -> np.ndarray:
        """[Q][k] u64 on the host, from the gathered block: ONE D2H copy into a pinned buffer allocated once.  The result is the caller's own
        array whatever Q is (copy=True, default); copy=False may return a VIEW of the pinned buffer -- it does where the shards are even (Q
        divisible by the rank count: the gathered block IS [Q][k]) -- which the next call overwrites."""
        import torch
        if getattr(self, "_host", None) is None:
            pin = self.all.is_cuda
            self._host = torch.empty(self.all.shape, dtype=torch.int64, pin_memory=pin)
        if self.all.is_cuda:
            self._host.copy_(self.all, non_blocking=True)
            torch.cuda.current_stream(self.all.device).synchronize()
        else:
            self._host.copy_(self.all)
        allv = self._host.numpy().view(np.uint64)
        if self.pad * self.world == self.Q:
            return allv.copy() if copy else allv
        out = np.empty((self.Q, self.k), dtype=np.uint64)
        for r in range(self.world):
            a, b = shard_range(self.Q, r, self.world)
            out[a:b] = allv[r * self.pad: r * self.pad + (b - a)]
        return out


def share_rows(eng, rank: int, world: int, n_nodes: int, group=None, slice_rows: int = 0) -> dict:
    """PEER ROWS (include/bang_c.h): after the load, before bang_alloc.  Every rank of the node keeps one slice of the adjacency rows in the
    HBM its index left over -- rank r the rows [r n, (r + 1) n), n = the smallest capacity among the ranks (or `slice_rows`) -- exports it
    (hipIpcGetMemHandle), the 64-byte handles travel through ONE all_gather_object of the job's process group, and every rank maps the W - 1
    others (hipIpcOpenMemHandle).  The search kernel then reads a parent's row from slice parent / n: its own HBM, a peer's over xGMI, or --
    beyond W n rows -- pinned host memory over PCIe as before.  Returns {"slice_rows", "rows_in_node_hbm", "fraction"}.

    Every rank executes the same three collectives whatever happens locally: a failure on one rank (no pull mode, out of HBM, an IPC handle
    that cannot be opened) travels WITH the collective, every rank then drops its slice table and raises the same error -- no rank is left
    waiting in a collective the others never reach."""
    import torch.distributed as dist

    def gather(x):
        out = [None] * world
        dist.all_gather_object(out, x, group=group)
        return out

    def first_error(items):
        errs = [f"rank {r}: {it['err']}" for r, it in enumerate(items) if it.get("err")]
        return errs[0] if errs else None

    # 1. capacities
    mine = {"err": None, "cap": 0}
    try:
        mine["cap"] = int(eng.rows_capacity())
    except Exception as ex:                                  # noqa: BLE001
        mine["err"] = repr(ex)[:200]
    caps = gather(mine)
    err = first_error(caps)
    n = 0
    if not err:
        n = int(slice_rows) if slice_rows else min(c["cap"] for c in caps)
        n = min(n, (n_nodes + world - 1) // world)          # (W n >= N: the whole graph sits in the node's HBM; no slice larger than needed)
    # 2. slice + export
    mine = {"err": err, "handle": bytes(64), "rows": 0}
    if not err and n > 0:
        try:
            first = min(rank * n, n_nodes)
            eng.rows_slice(first, max(0, min(n, n_nodes - first)))
            h, _, rows = eng.rows_export()
            mine["handle"], mine["rows"] = h, rows
        except Exception as ex:                              # noqa: BLE001
            mine["err"] = repr(ex)[:200]
    handles = gather(mine)
    err = first_error(handles)
    # 3. import, then agree on the outcome
    mine = {"err": err}
    if not err and n > 0:
        try:
            for r in range(world):
                if r == rank:
                    eng.rows_import(r, world, n, None)
                else:
                    eng.rows_import(r, world, n, handles[r]["handle"] if handles[r]["rows"] else bytes(64), rows=handles[r]["rows"])
        except Exception as ex:                              # noqa: BLE001
            mine["err"] = repr(ex)[:200]
    err = first_error(gather(mine))
    if err:
        try:                                                 # (host rows serve everything: results are the same)
            eng.rows_close_peers()
        except Exception:                                    # noqa: BLE001
            pass
        raise RuntimeError("peer rows: " + err)
    if n <= 0:
        return {"slice_rows": 0, "rows_in_node_hbm": 0, "fraction": 0.0}
    total = min(n_nodes, world * n)
    return {"slice_rows": n, "rows_in_node_hbm": total, "fraction": total / max(1, n_nodes)}


def unshare_rows(eng, group=None):
    """Tear-down of share_rows in two phases (after eng.free(), before eng.unload()): every rank closes its mappings of the other ranks'
    slices, the ranks meet, and only then may anyone free the slice the others had mapped."""
    import torch.distributed as dist
    eng.rows_close_peers()
    dist.barrier(group=group)
