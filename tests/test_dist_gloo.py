"""N > 1 path on CPU: two processes over gloo shard the query batch, search their shards (with the oracle
standing in for the GPU engine -- tests may use it) and rebuild the full result with the single all-gather
that bench.py issues over RCCL."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, Q, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
    import torch.distributed as dist
    from bang_amd import shard, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ix, q, _, _ = synth.make_index(1500, 32, "uint8", 32, 8, Q, K=10, n_clusters=16, seed=21, pq_iters=3)
    a, b = shard.shard_range(Q, rank, world)
    ids, _ = O.Oracle(ix).search(q[a:b], 10, 24, nthreads=1)
    full = shard.gather_ids(ids, Q, 10, rank, world)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full)
    # the same collective through the pre-allocated buffers of the device-side gather (CPU tensors stand in for device memory here;
    # on the GPU the engine writes `mine` itself: bang_query_dev_e)
    import torch
    dg = shard.DeviceGather(Q, 10, rank, world, torch.device("cpu"))
    dg.mine[: b - a] = torch.from_numpy(np.ascontiguousarray(ids).view(np.int64))
    dg.gather()
    assert np.array_equal(dg.local_ids(), ids)
    np.save(os.path.join(out_dir, f"dg{rank}.npy"), dg.batch_ids())
    if rank == 0:
        ref, _ = O.Oracle(ix).search(q, 10, 24, nthreads=2)
        np.save(os.path.join(out_dir, "ref.npy"), ref)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_reproduces_the_single_process_result(tmp_path):
    Q, world = 37, 2            # odd Q: ragged shards
    mp.spawn(_worker, args=(world, _free_port(), Q, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"r{r}.npy"), ref)
        assert np.array_equal(np.load(tmp_path / f"dg{r}.npy"), ref)


def test_shard_ranges_cover_the_batch():
    from bang_amd.shard import shard_range
    for Q in (1, 7, 10000):
        for W in (1, 2, 3, 8):
            cuts = [shard_range(Q, r, W) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == Q
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))


class _FakeEngine:
    """Stands in for bang_amd.Engine in the peer-rows exchange (no GPU here): records what share_rows asks of an engine."""

    def __init__(self, rank, capacity, fail_slice=False):
        self.rank, self.capacity, self.slice, self.imports, self.closed, self.fail_slice = rank, capacity, None, [], False, fail_slice

    def rows_capacity(self):
        return self.capacity

    def rows_slice(self, first, rows):
        if self.fail_slice:
            raise RuntimeError("out of device memory (test)")
        self.slice = (first, rows)

    def rows_export(self):
        first, rows = self.slice
        return (bytes([self.rank + 1]) * 64 if rows else bytes(64)), first, rows

    def rows_import(self, slot, n_slots, slice_rows, handle, rows=0):
        self.imports.append((slot, n_slots, slice_rows, handle, rows))

    def rows_close_peers(self):
        self.closed = True


def _peer_fail_worker(rank, world, port, out_dir):
    """rank 1 cannot place its slice: BOTH ranks must come out of share_rows with the same error (no rank left in a collective)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
    import torch.distributed as dist
    from bang_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _FakeEngine(rank, 400, fail_slice=(rank == 1))
    msg = "no error"
    try:
        shard.share_rows(eng, rank, world, 1000)
    except RuntimeError as ex:
        msg = str(ex)
    open(os.path.join(out_dir, f"f{rank}.txt"), "w").write(msg + "|" + str(eng.closed))
    dist.barrier()
    dist.destroy_process_group()


def _peer_worker(rank, world, port, N, caps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
    import json
    import torch.distributed as dist
    from bang_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _FakeEngine(rank, caps[rank])
    info = shard.share_rows(eng, rank, world, N)
    shard.unshare_rows(eng)
    json.dump({"info": info, "slice": eng.slice, "closed": eng.closed,
               "imports": [[s, w, n, (None if h is None else h[0]), rows] for s, w, n, h, rows in eng.imports]}, open(os.path.join(out_dir, f"p{rank}.json"), "w"))
    dist.destroy_process_group()


def test_peer_rows_exchange_two_ranks(tmp_path):
    """share_rows (the host logic of the peer-rows feature, include/bang_c.h bang_rows_*_e) over gloo with two ranks: the slice size is the
    smallest capacity among the ranks, rank r keeps the rows [r n, (r + 1) n), every rank imports its own slice (no handle) and the sibling's
    handle into the right slot; an index smaller than the capacities is split in halves (W n >= N: everything in the node's HBM)."""
    import json
    for N, caps, n_want in ((1000, (300, 400), 300), (1000, (5000, 7000), 500), (1001, (5000, 7000), 501)):
        d = tmp_path / f"n{N}_{caps[0]}"
        d.mkdir()
        mp.spawn(_peer_worker, args=(2, _free_port(), N, caps, str(d)), nprocs=2, join=True)
        for r in range(2):
            p = json.load(open(d / f"p{r}.json"))
            assert p["info"]["slice_rows"] == n_want and p["info"]["rows_in_node_hbm"] == min(N, 2 * n_want) and p["closed"] is True
            first = min(r * n_want, N)
            assert p["slice"] == [first, min(n_want, N - first)]
            assert sorted(i[0] for i in p["imports"]) == [0, 1] and all(i[1] == 2 and i[2] == n_want for i in p["imports"])
            for slot, _, _, h, rows in p["imports"]:
                assert (h is None) == (slot == r) and (h is None or h == slot + 1)          # the sibling's handle went into the sibling's slot
                if h is not None:                                                           # ... with the row count the sibling EXPORTED (the engine refuses a short slice)
                    assert rows == min(n_want, N - min(slot * n_want, N))


def test_peer_rows_failure_on_one_rank_reaches_every_rank(tmp_path):
    mp.spawn(_peer_fail_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg, closed = open(tmp_path / f"f{r}.txt").read().split("|")
        assert "rank 1" in msg and "out of device memory" in msg and closed == "True", (r, msg)
