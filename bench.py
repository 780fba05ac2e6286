#!/usr/bin/env python3
"""bench.py -- queries/sec @ 10-recall@10 >= 0.9 on a 10K-query batch (BASELINE.json metric), the roofline of the search
kernel and of the PQ-distance stage (K2) alone, and the CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sift1b_shape|sift1m|sift10m|deep100m_shape|small|tiny]
                    [--graph host|device|auto] [--pull -1|0|1] [--batches B] [--no-legs] [--legs k2,sift1m,...]

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE).

* For EVERY N the workload is `sift1b_shape` = BASELINE.json configs[3] (N = 1) / configs[4] (N > 1): the configuration the target
  number is quoted on (BANG_Base/test_driver.cpp:433-439, Cost_Analysis.pdf p.3) -- uint8, D = 128, R = 64, m = 70, 10 000 queries,
  L = 152, graph in host RAM.  The index is streamed through bang_load_stream_e (vectors + PQ codes into HBM, adjacency lists into
  256-byte pull rows in pinned host memory) and searched by ONE self-paced search-kernel launch per batch that pulls the rows over
  PCIe.  N is scaled to what the box's host memory holds (stated in config.workload).  `value`, `config`, `roofline` (that launch +
  `k2_alone` at m = 70) and `cpu_baseline` (the oracle on the same layout at reduced N, where the HIP engine is also checked
  against it) all belong to this configuration.
* N = 1 also carries the other single-GPU configurations as LEGS, each timed the same way on its own index, as nested objects
  `config.at_*` and -- because the driver's record keeps scalars only -- flattened into `config.<leg>_qps`, `_recall`, `_L`, `_frac`
  ...: `sift1m` (configs[1], recall-verified; default placement = HBM, plus `sift1m_host_pull`, `sift1m_host_walker`,
  `sift1m_L200`), `deep100m_shape` (configs[2]), `walker` (sift1b_shape with a resident graph image served by the C++ walker
  threads: the north-star data flow) and `sift10m` (a recall-verified structured index beyond the Infinity Cache).
* N > 1: the 10K-query batch is split into contiguous shards, one per rank; every rank searches its shard on its own replica of
  the PQ table and vectors, all ranks share ONE copy of the pull rows in host memory (BANG_PULL_ROWS_DIR), and ONE RCCL all-gather
  of the result ids -- straight from the device buffers bang_query_dev_e leaves them in -- ends the step ("scaling": "strong": the
  total work is fixed).  `--batches B` instead streams B whole 10K batches per rank and step ("scaling": "weak": throughput mode,
  no collective on the data path).

A "step" = one bang_query over the whole batch.  bang_init (visited-filter / worklist reset) is outside the timed bracket,
exactly as in the reference harness (BANG_Base/test_driver.cpp:432-439); the init-inclusive rate is reported beside it.  Every
step is bracketed by barrier + cuda.synchronize on both sides and the MAX over ranks is taken.

roofline.traffic (N = 1, default run): HBM bytes per launch of the search kernel, measured by two child runs of the primary
configuration under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE: separate passes) before this process touches the GPU -- live_traffic().
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
T_PROCESS_START = time.time()
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (N, D, dtype, R, m, Q, clusters)
    "sift1m": (1_000_000, 128, "uint8", 64, 32, 10_000, 256),   # BASELINE.json configs[1]
    "sift10m": (10_000_000, 128, "uint8", 64, 32, 10_000, 1024),  # structured index beyond the Infinity Cache (320 MB of codes, 3.9 GB graph):
                                                                  # Vamana-style build on the GPU (bang_amd/index_build.py), SURVEY 8 f-3
    "sift100m": (100_000_000, 128, "uint8", 64, 70, 10_000, 10_000),  # SIFT1B's layout (m = 70) on a recall-verified 1e8-point index: 38.8 GB
                                                                      # of graph entries, built on the GPU in about a minute
    "sift300m": (300_000_000, 128, "uint8", 64, 70, 10_000, 30_000),  # the same, three tenths of the headline's N: 116 GB of graph entries, 77 GB of pull rows;
                                                                      # built on the GPU in ~4 min by the sliced builder (peak 236 GiB of HBM; 4e8 would not fit the host beside its
                                                                      # 155 GB of graph entries and 102 GB of pull rows)
    "small": (100_000, 128, "uint8", 64, 32, 10_000, 64),       # quick functional run
    "tiny": (20_000, 128, "uint8", 64, 32, 1_000, 32),
}
ARITH_DTYPE = "f32"    # the path computes PQ sums and exact distances in float32 (u8/i8 subtract in int, accumulate in f32)


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the MI355X boxes expose 256
    hardware threads but grant a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


class Ctx:
    """Process-wide state of a bench run."""
    pass


# ---------------------------------------------------------------------------------------------------------- workloads
def shared_dir(ctx):
    """Where the ranks of one node share the host graph: a directory in /dev/shm (tmpfs = page cache), /tmp if that is too small."""
    tag = f"bang_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}"
    for base in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")):
        try:
            st = os.statvfs(base)
            if st.f_bavail * st.f_frsize > (1 << 30):
                return os.path.join(base, tag)
        except OSError:
            pass
    return os.path.join("/tmp", tag)


def build_workload(name, ctx, Q=0, shape_n=0, reserve_rows=True, stream=False, host_codes=False):
    """Returns a dict: ix, queries, gt_i, gt_d, d_codes, name, graph (natural placement), prefix (index files, N > 1), release().

    N > 1 (one process per GPU): ONE host graph for the node (the reference keeps one pIndex in host RAM,
    bang_search.cu:312-328).  Structured workloads: rank 0 builds the index and writes the reference-format FILES into a tmpfs
    directory; every rank loads them through bang_load, which maps `_disk.bin` shared and read-only.  Shape-only workloads: rank 0
    fills one shared mapping, the others map it; every rank generates the (identical, seeded) PQ codes on its own GPU."""
    import torch
    import torch.distributed as dist
    from bang_amd import formats, synth
    t0 = time.time()
    world, rank = ctx.world, ctx.rank
    prefix = None
    sdir = shared_dir(ctx) if world > 1 else None
    if sdir and rank == 0:
        os.makedirs(sdir, exist_ok=True)
    if name.endswith("_shape"):
        from tools import shape_workload
        shared = None
        stream = bool(stream and shape_workload.SHAPES[name]["graph"] == "host" and not host_codes)
        if world > 1:
            st = os.statvfs(os.path.dirname(sdir))
            n_plan = torch.tensor([shape_workload.plan_n(name, ctx.dev, shape_n, reserve_rows, stream, shared_bytes=st.f_bavail * st.f_frsize)
                                   if rank == 0 else 0], dtype=torch.int64, device=ctx.cdev)
            dist.broadcast(n_plan, 0)
            shape_n = int(n_plan.item())
            if not stream:                       # (a streamed index has no graph image to share: every rank runs the generator)
                shared = (os.path.join(sdir, f"{name}.graph"), rank == 0, dist.barrier)
        ix, queries, gt_i, gt_d, d_codes, wl_name, shape_graph = shape_workload.make(
            name, ctx.dev, n_override=shape_n, Q=Q or 10_000, log=log, shared=shared, reserve_rows=reserve_rows, stream=stream,
            planned=(world > 1), host_codes=host_codes)

        def rel():
            shape_workload.release(ix)
            if world > 1:
                dist.barrier()
                if rank == 0:
                    import shutil
                    shutil.rmtree(sdir, ignore_errors=True)      # the graph image and the engine's pull rows file
    else:
        N, D, dtype, R, m, Qd, ncl = WORKLOADS[name]
        wl_name = (f"{name}: SIFT1M-like structured synthetic, {dtype} N={N} D={D} R={R} m={m} "
                   f"Q={Q or Qd} k={ctx.k} (kNN+random-link graph, trained PQ, brute-force GT)")
        d_codes, shape_graph = None, "host"
        def make():
            if N > 2_000_000:                    # exact kNN by brute force stops being practical: partitioned search + robust prune
                from bang_amd import index_build
                kw = dict(select="groupmin", probes=12) if N > 20_000_000 else {}
                return index_build.make_index_large(N, D, dtype, R, m, Q or Qd, K=ctx.k, n_clusters=ncl, device=ctx.dev, log=log, **kw)
            return synth.make_index(N, D, dtype, R, m, Q or Qd, K=ctx.k, n_clusters=ncl, device=ctx.dev)
        if N > 2_000_000:
            wl_name = wl_name.replace("SIFT1M-like structured synthetic", "SIFT-like structured synthetic").replace(
                "kNN+random-link graph", "Vamana-style graph: robust-pruned (alpha 1.2) approximate kNN + reverse edges + small-world links")
        if world == 1:
            ix, queries, gt_i, gt_d = make()
            rel = lambda: None   # noqa: E731
        else:
            prefix = os.path.join(sdir, name)
            if rank == 0:
                ix0, q0_, gi0, gd0 = make()
                formats.write_index(prefix, ix0)
                np.save(prefix + "_queries.npy", q0_)
                np.save(prefix + "_gt_ids.npy", gi0)
                np.save(prefix + "_gt_dists.npy", gd0)
                del ix0
            dist.barrier()
            ix = formats.read_index(prefix, dtype, mmap_graph=True)       # host-side view for the oracle spot check; graph = shared map
            queries, gt_i, gt_d = np.load(prefix + "_queries.npy"), np.load(prefix + "_gt_ids.npy"), np.load(prefix + "_gt_dists.npy")

            def rel():
                dist.barrier()
                if rank == 0:
                    import shutil
                    shutil.rmtree(sdir, ignore_errors=True)
    torch.cuda.synchronize()
    log(f"[bench] workload built in {time.time() - t0:.1f}s: {wl_name}")
    return dict(ix=ix, queries=queries, gt_i=gt_i, gt_d=gt_d, d_codes=d_codes, name=wl_name, graph=shape_graph, release=rel,
                key=name, prefix=prefix, shared_dir=sdir)


# ---------------------------------------------------------------------------------------------------------- one measurement
def make_engine(wl, graph, ctx, lanes=0, threads=0, timing=1, pull=-1):
    """pull: host-graph placement only -- -1 = engine default (the kernel pulls adjacency rows over PCIe when the rows fit the host
    memory next to the graph, else the C++ walker serves them), 0 = walker, 1 = pull."""
    import bang_amd
    import torch.distributed as dist
    gm = {"host": bang_amd.GRAPH_HOST, "device": bang_amd.GRAPH_DEVICE, "auto": bang_amd.GRAPH_AUTO}[graph]
    eng = bang_amd.Engine(wl["ix"].dtype, graph=gm, device=ctx.local_rank, lanes=lanes, threads=threads, timing=timing, pull=pull)

    src = getattr(wl["ix"], "entry_source", None)
    if ctx.world > 1 and wl.get("shared_dir") and src is not None and not os.environ.get("BANG_BENCH_NO_VECTOR_BROADCAST"):
        # N > 1, streamed index: ONE rank reads the index.  Rank 0 streams it -- adjacency lists into the node's rows file, vectors
        # into a device buffer of its own -- and hands the vectors on from its HBM (a broadcast: RCCL over xGMI); the other ranks
        # map the rows file (signature = the hash rank 0 reports) and never touch an index entry (bang_load_shared_e).
        import ctypes as C
        import torch
        ix = wl["ix"]
        vb = ix.D * (4 if ix.dtype == "float" else 1)
        os.environ["BANG_PULL_ROWS_DIR"] = wl["shared_dir"]
        vec = torch.empty(ix.N * vb + 256, dtype=torch.uint8, device=ctx.dev)
        h = torch.zeros(1, dtype=torch.int64, device=ctx.cdev)
        t0 = time.time()
        status = torch.zeros(1, dtype=torch.int64, device=ctx.cdev)       # rank 0's load may fail (rows file does not fit tmpfs ...): every rank
        err = None                                                         # learns it from this word and raises, instead of hanging in the broadcast
        if ctx.rank == 0:
            try:
                eng.load_stream(ix, src[0], C.byref(src[1]), d_codes=wl["d_codes"], code_stride=getattr(ix, "code_stride", 0), d_vectors=vec.data_ptr())
                hv_ = eng.rows_hash()
                h[0] = hv_ - (1 << 64) if hv_ >= (1 << 63) else hv_      # (u64 -> the int64 the collective carries)
            except Exception as ex:                                        # noqa: BLE001
                err, status[0] = ex, 1
        dist.broadcast(status, 0)
        if int(status.item()) != 0:
            raise RuntimeError(f"rank 0 could not load the index: {err}" if err else "rank 0 could not load the index (see its log)")
        dist.broadcast(h, 0)
        t1 = time.time()
        if ctx.cdev == ctx.dev:
            step_b = 1 << 32                     # (4 GB per call: a 121 GB count in one collective has never been exercised here)
            for a in range(0, vec.numel(), step_b):
                dist.broadcast(vec[a:a + step_b], 0)
        else:                                    # gloo dry runs: through the host
            hv = vec.cpu() if ctx.rank == 0 else torch.empty(vec.shape, dtype=torch.uint8)
            dist.broadcast(hv, 0)
            if ctx.rank != 0:
                vec.copy_(hv)
        torch.cuda.synchronize()
        if ctx.rank != 0:
            eng.load_shared(ix, vec.data_ptr(), int(h.item()) & ((1 << 64) - 1), d_codes=wl["d_codes"], code_stride=getattr(ix, "code_stride", 0))
        wl["_vectors"] = vec                     # (the engines read it until bang_unload)
        log(f"[bench] rank 0 streamed the index in {t1 - t0:.1f}s, vectors broadcast to {ctx.world - 1} rank(s) in {time.time() - t1:.1f}s")
        return eng

    def load():
        if src is not None:                      # streamed shape index: the engine pulls the generator's entries through in chunks
            import ctypes as C
            eng.load_stream(wl["ix"], src[0], C.byref(src[1]), d_codes=wl["d_codes"], code_stride=getattr(wl["ix"], "code_stride", 0))
        elif wl.get("prefix"):
            eng.load(wl["prefix"])               # bang_load on the shared index files (`_disk.bin` streamed or mapped, never copied)
        else:
            eng.load_index(wl["ix"], d_codes=wl["d_codes"], code_stride=(getattr(wl["ix"], "code_stride", 0) if wl["d_codes"] else 0))
    if ctx.world > 1 and wl.get("shared_dir"):
        # one copy of the pull rows per node: rank 0 builds the rows file in the shared directory, the others map it
        os.environ["BANG_PULL_ROWS_DIR"] = wl["shared_dir"]
        import torch
        status = torch.zeros(1, dtype=torch.int64, device=ctx.cdev)
        err = None
        if ctx.rank == 0:
            try:
                load()
            except Exception as ex:                                        # noqa: BLE001  (the other ranks must not wait for a barrier that never comes)
                err, status[0] = ex, 1
        dist.broadcast(status, 0)
        if int(status.item()) != 0:
            raise RuntimeError(f"rank 0 could not load the index: {err}" if err else "rank 0 could not load the index (see its log)")
        if ctx.rank != 0:
            load()
    else:
        load()
    return eng


def run_once(eng, my_q, ctx, timed=False, gather=True):
    """One step: bang_init (untimed), then bang_query over this rank's shard and -- N > 1, strong scaling -- the single collective
    of the job.  With RCCL the shard's ids stay in device memory (bang_query_dev_e) and are all-gathered from there; rank 0 copies
    the gathered [Q][k] block to the host once.  Returns (ids of this rank, dists of this rank or None, seconds, search s, gather s)."""
    import torch
    import torch.distributed as dist
    from bang_amd import shard
    t_i = time.perf_counter()
    eng.init(my_q.shape[0])
    ctx.last_init_s = time.perf_counter() - t_i          # bang_init alone (it returns when the device is done with it)
    if timed:
        if ctx.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    collective = (ctx.world > 1 or getattr(ctx, "force_gather", False)) and gather and not ctx.weak
    dg = getattr(ctx, "dgather", None) if collective else None
    t_a = time.perf_counter()
    if dg is not None:
        eng.query_dev(my_q, dg.mine.data_ptr(), dg.dists.data_ptr())
        t_b = time.perf_counter()
        dg.gather()
        if ctx.rank == 0:
            ctx.batch_ids = dg.batch_ids()               # the batch's answer reaches the host on one rank (one D2H copy)
        else:
            torch.cuda.synchronize()
        ids, dists = None, None
    else:
        ids, dists = eng.query(my_q)
        t_b = time.perf_counter()
        if collective:                                   # (gloo dry runs: host buffers)
            ctx.batch_ids = shard.gather_ids(ids, ctx.Q_total, ctx.k, ctx.rank, ctx.world, device=ctx.cdev)
    t_c = time.perf_counter()
    if timed:
        torch.cuda.synchronize()
        if ctx.world > 1:
            dist.barrier()
    t_d = time.perf_counter()
    if dg is not None:
        ids, dists = dg.local_ids(), dg.local_dists()    # (untimed: this rank's block for the recall / parity / property checks)
    return ids, dists, t_d - t_a, t_b - t_a, t_c - t_b


def check_properties(ix, my_q, ids, dists, k):
    """Size-independent result properties (shape-only workloads keep their PQ codes only in HBM, so the oracle cannot run):
    ids in range and distinct, distances ascending and equal to the exact distance of the returned id recomputed on the host."""
    ok = True
    isz = 4 if ix.dtype == "float" else 1
    npd = np.float32 if ix.dtype == "float" else np.uint8
    Qr = my_q.shape[0]
    for qi in range(0, Qr, max(1, Qr // 256)):
        row = ids[qi].astype(np.int64)
        if not (row < ix.N).all():
            return False
        vec = np.ascontiguousarray(ix.graph[row, : ix.D * isz]).view(npd).reshape(k, ix.D).astype(np.float64)
        ex = ((vec - my_q[qi].astype(np.float64)) ** 2).sum(axis=1)
        dd = dists[:, qi].astype(np.float64)
        ok &= bool(len(set(row.tolist())) == k and (np.diff(dd) >= 0).all() and np.allclose(dd, ex, rtol=1e-5))
    return ok


def measure(eng, wl, my_q, L, steps, warmup, ctx, graph, traffic_key=None, batches=1):
    """Times `steps` steps of `batches` bang_query calls each at worklist length L on an allocated engine.  Returns a dict with
    the rate, the per-step times, the engine statistics and the roofline of the search kernel."""
    import torch
    import torch.distributed as dist
    ix, k = wl["ix"], ctx.k
    Qr = my_q.shape[0]
    for _ in range(warmup):
        run_once(eng, my_q, ctx, timed=True)
    step_s, init_s, search_s, gather_s = [], [], [], []
    keys_max = ("iterations", "persistent", "vectors_on_device", "graph_mode", "lanes", "walker_threads", "wg_queries",
                "workgroups", "hops_p50", "hops_p99", "hops_max", "graph_pull", "code_stride", "rows_in_hbm")
    agg = dict(front_ms=0.0, front_busy_ms=0.0, walker_ms=0.0, sync_ms=0.0, enqueue_ms=0.0, dist_evals=0, front_launches=0,
               fetched=0, candidates=0, h2d_bytes=0, pulled_bytes=0, filter_loads_skipped=0)
    agg.update({kk: 0 for kk in keys_max})
    ids = dists = None
    for _ in range(steps):
        t_init = 0.0
        el = 0.0
        e_s = e_g = 0.0
        for b in range(batches):
            ids, dists, e1, es1, eg1 = run_once(eng, my_q, ctx, timed=True, gather=(batches == 1))
            el += e1
            e_s += es1
            e_g += eg1
            t_init += ctx.last_init_s
            st = eng.stats()
            for key in agg:
                agg[key] = max(agg[key], st[key]) if key in keys_max else agg[key] + st[key]
        init_s.append(el + t_init)                       # bang_init + bang_query, nothing else (round 3 also counted the statistics read-back)
        step_s.append(el)
        search_s.append(e_s)
        gather_s.append(e_g)
    times = torch.tensor([step_s, init_s, search_s, gather_s], dtype=torch.float64, device=ctx.cdev)
    if ctx.world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    total = float(times[0].sum().item())
    if ctx.world == 1:
        n_q = Qr * batches
    else:                                               # sharded: the ranks' shards add up to the batch; weak: a batch per rank
        n_q = ctx.Q_total if batches == 1 else Qr * ctx.world * batches       # queries all ranks processed per step
    res = dict(L=L, queries_per_s=round(n_q * steps / total, 1), ms_per_step=round(1e3 * total / steps, 4),
               qps_incl_init=round(n_q * steps / float(times[1].sum().item()), 1),
               step_ms=[round(1e3 * float(t), 2) for t in times[0].tolist()][:64], ids=ids, dists=dists, agg=agg,
               search_ms=round(1e3 * float(times[2].sum().item()) / steps, 4),      # bang_query of the slowest rank, mean over the steps
               gather_ms=round(1e3 * float(times[3].sum().item()) / steps, 4))      # the collective (+ rank 0's copy of the batch to the host)
    # ---- roofline of the search kernel of this measurement
    m = ix.m
    bpe = m + 8                                         # m code bytes + 4 B id + 4 B distance per evaluation
    roof = None
    launches = agg["front_launches"]
    if launches and agg["front_ms"] > 0:
        persistent = bool(agg["persistent"])
        evals_per_launch = agg["dist_evals"] / launches
        avg_ms = (agg["front_busy_ms"] if persistent else agg["front_ms"]) / launches
        achieved = evals_per_launch * bpe / (avg_ms * 1e-3) / 1e9
        traffic, traffic_note, by_stream = None, None, None
        tf = os.path.join(ROOT, "profiles", f"traffic_{traffic_key}.json") if traffic_key else None
        live = getattr(ctx, "live_traffic", None) if traffic_key and traffic_key == getattr(ctx, "live_traffic_key", None) else None
        if live and live.get("bytes") and persistent:
            traffic, traffic_note = live["bytes"], live["note"]
            # where the bytes go, from the engine's own counts: one 128-byte line per code row (rows 128 B apart; 1.6 lines of a packed
            # 70-byte row), 256 B per adjacency row read from the HBM copy, the rest of the reads = filter words (one 128-byte line
            # each, less what L2 served), the writes = filter stores (32 B each)
            cs = int(agg.get("code_stride") or m)
            rows_b = evals_per_launch * (128.0 if cs >= 128 else 128.0 * (1.0 + (m - 1) / 128.0))
            adj_b = max(0.0, (agg["candidates"] / launches - Qr) * 256.0 - (agg["pulled_bytes"] / launches if graph == "host" else 0.0))
            by_stream = {"code_rows": int(rows_b), "adjacency_rows_from_hbm": int(adj_b),
                         "filter_reads": int(max(0.0, live["hbm_read"] - rows_b - adj_b)), "filter_writes": int(live["hbm_write"]),
                         "pcie_adjacency_rows": int(live.get("pcie_read", 0))}
        elif tf and os.path.exists(tf) and ctx.world == 1:
            try:
                tj = json.load(open(tf))
                traffic = tj.get("search_kernel_hbm_bytes_per_launch" if persistent else "front_kernel_hbm_bytes_per_launch")
                traffic_note = (f"HBM bytes per launch from the committed rocprofv3 PMC passes of this command "
                                f"(profiles/traffic_{traffic_key}.json) -- NOT re-measured in this run"
                                + (f" ({live['note']})" if live and not live.get("bytes") else ""))
            except Exception:
                traffic = None
        # scalars first (the driver's record keeps the leading scalars of an object), prose and nested objects behind them
        roof = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                "traffic_over_algorithmic": (round(traffic / (evals_per_launch * bpe), 3) if traffic else None),
                # what the launch MOVES (every read request a 128-byte line, 32 B per scattered store) against the same peak: the algorithmic
                # fraction above counts m + 8 bytes per evaluation, the memory system carries ~4x that
                "hbm_traffic_GBps": (round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None),
                "hbm_traffic_frac": (round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None),
                "k2_alone_frac": None, "k2_alone_GBps": None,
                "algorithmic_bytes_per_launch": round(evals_per_launch * bpe, 1),
                "avg_launch_us": round(avg_ms * 1e3, 3), "launches": launches, "bytes_per_distance_eval": bpe,
                "kernel": ("search kernel, ONE launch per batch (K5 filter + K2 PQ distance + K4 parent + K3a sort + K3b merge for every "
                           "iteration of every query)") if persistent else "front_kernel (K5 filter + K2 PQ distance + K4 parent, fused)",
                "timer": "in-kernel s_memrealtime stamps (100 MHz) on every launch of the timed steps; cross-checked against "
                         "rocprofv3 --kernel-trace in profiles/"}
        if traffic_note:
            roof["traffic_note"] = traffic_note
        if by_stream:
            roof["traffic_by_stream"] = by_stream
        if persistent and graph == "host" and agg.get("graph_pull"):
            pb = agg["pulled_bytes"] / launches
            roof["pcie_pull"] = {"bytes_per_launch": int(pb), "achieved_GBps": round(pb / (avg_ms * 1e-3) / 1e9, 2),
                                 "note": "256-byte adjacency rows the kernel reads from pinned host memory; 57 GB/s of such rows measured "
                                         "(tools/dev/gpu_pull_bench.hip)"}
        elif persistent and graph == "host":
            h2d = agg["h2d_bytes"] / launches
            roof["pcie_h2d"] = {"bytes_per_launch": int(h2d), "achieved_GBps": round(h2d / (avg_ms * 1e-3) / 1e9, 2),
                                "note": "adjacency rows (+ vectors if shipped) the walker threads store through the PCIe BAR while the "
                                        "kernel runs; 64-B write TLPs carry at most ~46-48 GB/s on x16 Gen5"}
    res["roofline"] = roof
    return res


def host_loop_name(a, graph):
    if not a["persistent"]:
        return "launch per iteration"
    if graph == "device":
        return "search kernel, self-paced (graph in HBM)"
    if a.get("graph_pull"):
        return "search kernel, self-paced: adjacency rows pulled from pinned host memory over PCIe by the kernel (no walker thread)"
    return "search kernel, host-paced: C++ walker threads write adjacency rows through the PCIe BAR"


def leg_summary(res, wl, graph, recall=None, props=None, extra=None):
    a = res["agg"]
    out = {"workload": wl["name"], "graph": graph, "L": res["L"], "queries_per_s": res["queries_per_s"],
           "ms_per_batch": res["ms_per_step"], "iterations": a["iterations"],
           "hops_p50_p99_max": [a["hops_p50"], a["hops_p99"], a["hops_max"]],
           "host_loop": host_loop_name(a, graph),
           "rerank_vectors": ("graph entries in HBM" if graph == "device" else
                              "packed copy in HBM" if a["vectors_on_device"] else "shipped by the walker (PCIe)"),
           "steps": len(res["step_ms"]), "step_ms_min_max": [min(res["step_ms"]), max(res["step_ms"])]}
    if a.get("graph_pull"):
        out["pcie_pulled_bytes_per_batch"] = int(a["pulled_bytes"] // max(1, len(res["step_ms"])))
    if recall is not None:
        out["recall_at_10"] = round(recall, 3)
    if props is not None:
        out["result_properties_ok"] = props
    if res["roofline"]:
        r = res["roofline"]
        out["roofline"] = {kk: r[kk] for kk in ("achieved", "frac", "avg_launch_us", "algorithmic_bytes_per_launch", "traffic") if kk in r}
    if extra:
        out.update(extra)
    return out


# ---------------------------------------------------------------------------------------------------------- K2 alone
def k2_alone(D, m, dtype, ctx, table_bytes=4 << 30, rows_per_launch=40_000_000, reps=5, stride=0):
    """The PQ-distance stage (K2, compute_neighborDist_par, bang_search.cu:1201-1241) ALONE: `bang_k_pqdist` over enough
    (query, neighbour) pairs that one launch takes >= 1 ms, on a random code table far larger than the 256 MB Infinity Cache.
    Timed with HIP events on the launch stream.  Algorithmic bytes = evaluations x (m + 8)."""
    import torch
    from bang_amd import binding as B
    from bang_amd.synth import chunk_offsets
    dev = ctx.dev
    rb = stride or m                                    # bytes between rows (stride > m: padded rows, e.g. 128 for m = 70)
    N = int(table_bytes // rb)
    Qk = rows_per_launch // 64
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    codes = torch.empty(N * rb + 256, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for a in range(0, N * rb, step):
        b = min(N * rb, a + step)
        codes[a:b] = torch.randint(0, 256, (b - a,), dtype=torch.uint8, device=dev, generator=g)
    codes[N * rb:] = 0
    chunk_off = chunk_offsets(D, m)
    psz, mp = B.pq_layout(chunk_off, D, m)
    if psz == 0:
        return None
    rng = np.random.default_rng(3)
    pivots = (rng.standard_normal((256, D)) * 30).astype(np.float32)
    nhi, table = B.pack_pivots_ragged(pivots, chunk_off, D, m, mp) if psz == 2 else (0, None)
    if nhi and not B.lib().bang_ragged_supported(psz, mp, nhi, m):
        nhi = 0
    packed = torch.from_numpy(table if nhi else B.pack_pivots(pivots, chunk_off, D, m, psz, mp).reshape(-1)).to(dev)
    nbrs = torch.zeros((Qk, B.NBR_STRIDE), dtype=torch.int32, device=dev)
    nbrs[:, :64] = torch.randint(0, N, (Qk, 64), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    dist_o = torch.zeros((Qk, B.NBR_STRIDE), dtype=torch.float32, device=dev)
    cnt = torch.full((Qk,), 64, dtype=torch.int32, device=dev)
    # the neighbour rows belong to 10 000 distinct queries (row q -> query q mod 10 000): a search evaluates every query against
    # a new neighbour row per iteration, it does not meet 625 000 different queries
    n_queries = 10_000
    qc = torch.randn((n_queries, mp * psz), dtype=torch.float32, device=dev, generator=g)
    seed = torch.zeros(80, dtype=torch.int32, device=dev)
    p = B.IterParams()
    p.Q, p.R, p.m, p.L, p.medoid, p.iter, p.first = Qk, 64, m, 16, 0, 2, 0
    p.n_all = n_queries
    p.psz, p.mp, p.pq_nhi = psz, mp, nhi
    p.code_stride = stride
    p.d_codes, p.d_pivots_packed, p.d_qc = codes.data_ptr(), packed.data_ptr(), qc.data_ptr()
    p.d_nbrs, p.d_dist, p.d_cnt, p.d_seed = nbrs.data_ptr(), dist_o.data_ptr(), cnt.data_ptr(), seed.data_ptr()
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    entry = B.lib().bang_k_pqdist_stream          # K2 alone, streaming form (next row in flight while the current one is reduced)
    for _ in range(6):                                  # (the first launches on a freshly written table run ~10 % slower)
        B._check(entry(C.byref(p), sp), "bang_k_pqdist_stream")
    torch.cuda.synchronize()
    us = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        B._check(entry(C.byref(p), sp), "bang_k_pqdist_stream")
        e1.record(stream)
        e1.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
    avg = float(np.mean(us))
    evals = Qk * 64
    ach = evals * (m + 8) / (avg * 1e-6) / 1e9
    out = {"kernel": "pqdist_stream_kernel (K2 alone) via bang_k_pqdist_stream", "m": m, "D": D, "psz_mp": [psz, mp],
           "neighbour_rows": Qk, "distinct_queries": n_queries,
           "code_stride": rb, "code_table_bytes": N * rb, "evals_per_launch": evals, "algorithmic_bytes_per_launch": evals * (m + 8),
           "avg_launch_us": round(avg, 1), "min_launch_us": round(min(us), 1), "achieved": round(ach, 1), "unit": "GB/s",
           "peak": HBM_PEAK_GBPS, "frac": round(ach / HBM_PEAK_GBPS, 4), "rows_per_s": round(evals / (avg * 1e-6) / 1e9, 2),
           "timer": "HIP events on the launch stream"}
    del codes, nbrs, dist_o, cnt, qc, packed
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------- one configuration
def batch_recall(O, gt_i, gt_d, ids, k, q0, q1, ctx):
    """10-recall@10 of the WHOLE batch: the ranks' shard recalls weighted by their shard sizes (== the single-process number;
    a MIN over shards would make the L sweep depend on the rank count)."""
    import torch
    import torch.distributed as dist
    if gt_i is None:
        return float("nan")
    r = O.recall(gt_i[q0:q1], gt_d[q0:q1], ids, k)
    if ctx.world == 1:
        return r
    t = torch.tensor([r * (q1 - q0), float(q1 - q0)], dtype=torch.float64, device=ctx.cdev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0].item() / t[1].item())


def run_config(name, ctx, args, O, *, graph="", pull=-1, L=0, steps=5, warmup=1, stream=True, reserve_rows=True, Q=0, shape_n=0,
               traffic=True, batches=1, lanes=0, threads=0, keep=False, host_codes=False):
    """Builds workload `name`, loads an engine in the requested placement, chooses L (structured workloads: the smallest L on the
    harness grid k, k+12, ... with 10-recall@10 >= target; shape-only workloads: 152, the reference's SIFT1B setting), times
    `steps` steps and checks the results (structured: first 64 queries of this rank against the oracle; shape-only: the
    size-independent result properties).  Returns a dict; with keep=True the workload / engine stay alive (caller releases)."""
    import torch
    from bang_amd import shard
    k = ctx.k
    wl = build_workload(name, ctx, Q=Q, shape_n=shape_n, reserve_rows=reserve_rows, stream=stream, host_codes=host_codes)
    ix, queries, gt_i, gt_d = wl["ix"], wl["queries"], wl["gt_i"], wl["gt_d"]
    graph = graph or ("auto" if gt_i is not None else wl["graph"])
    Qt = queries.shape[0]
    ctx.Q_total = Qt
    weak = ctx.world > 1 and batches > 1
    ctx.weak = weak
    q0, q1 = (0, Qt) if weak else shard.shard_range(Qt, ctx.rank, ctx.world)
    my_q = np.ascontiguousarray(queries[q0:q1])
    Qr = q1 - q0
    eng = make_engine(wl, graph, ctx, lanes=lanes, threads=threads, timing=0 if args.no_events else 1, pull=pull)
    # N > 1: the shard's ids stay in device memory (bang_query_dev_e) until the collective (BANG_BENCH_HOST_GATHER=1: the r02 host bounce)
    ctx.dgather = None
    if (ctx.world > 1 or getattr(ctx, "force_gather", False)) and not weak and not os.environ.get("BANG_BENCH_HOST_GATHER"):
        ctx.dgather = shard.DeviceGather(Qt, k, ctx.rank, ctx.world, ctx.dev, coll_device=ctx.cdev)
    recall = float("nan")
    if L == 0 and gt_i is not None:
        for cand in range(k, 513, 12):                   # the harness's sweep grid, test_driver.cpp:376-417
            eng.set_searchparams(k, cand)
            eng.alloc(Qr)
            ids = run_once(eng, my_q, ctx)[0]
            eng.free()
            r = batch_recall(O, gt_i, gt_d, ids, k, q0, q1, ctx)
            log(f"[bench] {name} L={cand:3d} recall={r:.2f}")
            if r >= args.recall_target:
                L, recall = cand, r
                break
        if L == 0:
            raise RuntimeError("recall target not reached on the L grid")
    elif L == 0:
        L = 152                                          # reference's SIFT1B setting, BANG_Inmemory/parANN.h:99
    eng.set_searchparams(k, L)
    eng.alloc(Qr)
    placement_note = None
    if graph == "auto":                                  # what did "auto" resolve to?
        run_once(eng, my_q, ctx)
        graph = "device" if eng.stats()["graph_mode"] == 1 else "host"
        placement_note = f"auto -> {graph} (engine default: graph in HBM when it fits next to the PQ codes with 16 GB to spare)"
    if getattr(ctx, "live_primary", False):              # the configuration main()'s live PMC passes measured
        ctx.live_traffic_key, ctx.live_primary = f"{name}_{graph}", False
    res = measure(eng, wl, my_q, L, steps, warmup, ctx, graph, traffic_key=(f"{name}_{graph}" if traffic else None),
                  batches=batches if weak else 1)
    orc = None
    if gt_i is not None:
        recall = batch_recall(O, gt_i, gt_d, res["ids"], k, q0, q1, ctx)
        orc = O.Oracle(ix)
        chk = min(64, Qr)
        ids_o, _ = orc.search(my_q[:chk], k, L)
        ok = bool(np.array_equal(res["ids"][:chk], ids_o))
    else:
        ok = check_properties(ix, my_q, res["ids"], res["dists"], k)
    gathered_ok = None
    if (ctx.world > 1 or getattr(ctx, "force_gather", False)) and not weak and ctx.rank == 0 and getattr(ctx, "batch_ids", None) is not None:
        # what the collective delivered, against the oracle over the WHOLE batch (where the oracle can run: host-side PQ codes)
        if orc is None and getattr(ix, "codes", None) is not None and ix.codes.shape[0] == ix.N:
            orc = O.Oracle(ix)
        if orc is not None and Qt <= 20_000:
            ids_all, _ = orc.search(queries, k, L, nthreads=usable_cpus())
            gathered_ok = bool(np.array_equal(ctx.batch_ids, ids_all))
    out = dict(wl=wl, eng=eng, res=res, L=L, recall=recall, ok=ok, graph=graph, orc=orc, my_q=my_q, q0=q0, q1=q1, Qr=Qr,
               placement_note=placement_note, structured=gt_i is not None, name=name, gathered_ok=gathered_ok)
    if not keep:
        release_config(out)
    return out


def release_config(rc):
    import torch
    if rc.get("eng") is not None:
        e = rc["eng"]
        e.free(); e.unload(); e.close()
        rc["eng"] = None
    if rc.get("wl") is not None:
        rc["wl"]["release"]()
        rc["wl"] = None
    rc["orc"] = None
    torch.cuda.empty_cache()


def flat(cfg, prefix, leg):
    """The key facts of a leg as SCALARS of `config` (the driver's record keeps scalars only; the nested leg stays beside them)."""
    for kk, name in (("queries_per_s", "qps"), ("ms_per_batch", "ms"), ("L", "L"), ("recall_at_10", "recall"),
                     ("parity_vs_oracle_first_64", "parity_ok"), ("result_properties_ok", "props_ok")):
        if kk in leg and leg[kk] is not None:
            cfg[f"{prefix}_{name}"] = leg[kk]
    if isinstance(leg.get("roofline"), dict):
        cfg[f"{prefix}_frac"] = leg["roofline"].get("frac")
    if "hops_p50_p99_max" in leg:
        cfg[f"{prefix}_hops_p50"], cfg[f"{prefix}_hops_p99"] = leg["hops_p50_p99_max"][0], leg["hops_p50_p99_max"][1]


# ---------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline_structured(rc, O, k, queries):
    nthreads = usable_cpus()
    orc, L, Q = rc["orc"], rc["L"], queries.shape[0]
    orc.search(queries[: min(Q, 512)], k, L, nthreads=nthreads)     # warm
    reps, t_cpu = 0, 0.0
    while reps < 5 and t_cpu < 10.0:
        t_a = time.perf_counter()
        orc.search(queries, k, L, nthreads=nthreads)
        t_cpu += time.perf_counter() - t_a
        reps += 1
    return {"value": round(Q * reps / t_cpu, 1), "unit": "queries/s", "cores": nthreads, "kind": "port",
            "sample": f"{reps} x the full {Q}-query batch at L={L} through oracle/ (C + OpenMP, {nthreads} threads = "
                      f"the CPU quota of this box; {os.cpu_count()} hardware threads visible), same timed region (search only)"}


def cpu_baseline_shape(name, ctx, args, O, L):
    """The oracle beside a shape-only workload: the same layout (dtype, D, R, m, L, iteration cap, ~56 evaluations per
    iteration) on an index of REDUCED N whose PQ codes also exist in host memory (the full-size index keeps them only in HBM).
    Per-query work does not depend on N once the tables are far larger than the CPU caches (N = 2e7: 1.4 GB of codes, 7.8 GB of
    graph entries vs 2 x 384 MB of L3).  The HIP engine runs the SAME reduced index first: its ids must equal the oracle's on
    every sampled query, which puts an oracle-checked run of this layout into every bench run."""
    from tools import shape_workload
    k = ctx.k
    n_small = int(os.environ.get("BANG_CPU_BASELINE_N", "20000000"))
    t0 = time.time()
    ix, queries, _, _, _, wl_name, _ = shape_workload.make(name, ctx.dev, n_override=n_small, Q=10_000, log=log, host_codes=True, planned=True)
    wl = dict(ix=ix, queries=queries, gt_i=None, gt_d=None, d_codes=None, name=wl_name, graph="host", prefix=None, shared_dir=None)
    eng = make_engine(wl, "host", ctx, timing=0)
    eng.set_searchparams(k, L)
    eng.alloc(queries.shape[0])
    ids_g = run_once(eng, queries, ctx)[0]
    eng.free(); eng.unload(); eng.close()
    nthreads = usable_cpus()
    orc = O.Oracle(ix)
    orc.search(queries[:256], k, L, nthreads=nthreads)               # warm
    # the whole 10 K batch, again and again until ~10 s of CPU work are on the clock (the first pass also checks the HIP engine's ids)
    done, t_cpu, ids_o, reps = queries.shape[0], 0.0, None, 0
    while t_cpu < 10.0 and reps < 12:
        t_a = time.perf_counter()
        ids_r, _ = orc.search(queries, k, L, nthreads=nthreads)
        t_cpu += time.perf_counter() - t_a
        reps += 1
        if ids_o is None:
            ids_o = ids_r
    parity = bool(np.array_equal(ids_g, ids_o))
    shape_workload.release(ix)
    log(f"[bench] cpu baseline ({name}, N={ix.N}): {reps} x {done} queries in {t_cpu:.1f}s on {nthreads} threads, parity with the HIP engine: {parity} "
        f"({time.time() - t0:.0f}s in all)")
    return {"value": round(done * reps / t_cpu, 1), "unit": "queries/s", "cores": nthreads, "kind": "port",
            "sample": f"{reps} x the {done}-query batch ({t_cpu:.0f} s of CPU work) of the {name} layout (m={ix.m}, L={L}, iteration cap L+49) at reduced N={ix.N} with the PQ codes in "
                      f"host memory, through oracle/ (C + OpenMP, {nthreads} threads = the CPU quota of this box; {os.cpu_count()} hardware "
                      f"threads visible), search only; per-query work is N-independent once the tables exceed the caches",
            "hip_ids_equal_oracle_on_sample": parity}


# ---------------------------------------------------------------------------------------------------------- HBM traffic, live
def live_traffic(args, log):
    """roofline.traffic measured in THIS run: two child runs of this very command (primary workload only, 3 timed steps) under
    `rocprofv3 --pmc` -- separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes:
        pass A: FETCH_SIZE + TCC_EA0_RDREQ_DRAM_32B_sum      pass B: WRITE_SIZE + TCC_EA0_RDREQ_IO_32B_sum + TCC_EA0_WRREQ_sum
    Calibration on known byte counts in this path's access shapes (tools/traffic_calib.hip, profiles/r04_traffic_calibration.md):
    EVERY read request of gfx950's L2 to memory is a 128-byte line -- a 4-byte filter probe as much as a code row or a streamed read --
    and FETCH_SIZE tallies each at 64 bytes (exactly half, for every shape: the guide's x2 holds throughout), while
    TCC_EA0_RDREQ_DRAM_32B x 32 and WRITE_SIZE x 1024 (32 bytes per scattered 4-byte store) are byte-exact.  `bytes` = HBM reads
    (DRAM_32B x 32) + HBM writes (WRITE_SIZE x 1024); reads over PCIe (the pulled adjacency rows, IO_32B x 32) are listed apart.
    The children run and exit BEFORE this process initialises the GPU (they need the HBM the parent would hold).
    Returns {bytes per launch, parts, note} or {None, why}."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"bytes": None, "note": "rocprofv3 is not on PATH"}
    steps = 3
    # the program behind `--` is the interpreter itself (no PATH look-up, no shim, no `env` hop: the profiler's preloaded library
    # initialises the GPU before the program starts, and an exec from such a process takes the machine down on this pool)
    child = [os.path.realpath(sys.executable), os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--steps", str(steps), "--warmup", "1", "--no-legs",
             "--no-cpu-baseline", "--pull", str(args.pull)]
    for flag, val in (("--graph", args.graph), ("--L", args.L), ("--queries", args.queries), ("--shape-n", args.shape_n),
                      ("--lanes", args.lanes), ("--threads", args.threads)):
        if val:
            child += [flag, str(val)]
    if args.resident_graph:
        child.append("--resident-graph")
    env = dict(os.environ, BANG_BENCH_NO_TRAFFIC="1", TMPDIR="/tmp")
    got = {}
    passes = (("A", ("FETCH_SIZE", "TCC_EA0_RDREQ_DRAM_32B_sum")), ("B", ("WRITE_SIZE", "TCC_EA0_RDREQ_IO_32B_sum", "TCC_EA0_WRREQ_sum")))
    for tag, counters in passes:
        d = tempfile.mkdtemp(prefix="bang_pmc_", dir="/tmp")
        t0 = time.time()
        try:
            pr = subprocess.Popen(["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child,
                                  cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                so, _ = pr.communicate(timeout=120)                 # (a pass takes ~30 s; the run must stay within minutes whatever the profiler does)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)            # (the session this call started: nothing else is in it)
                pr.communicate()
                return {"bytes": None, "note": f"PMC pass {tag} did not finish in 120 s"}
            f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            rows = [r for r in csv.DictReader(open(f[0]))] if f else []
            sel = [r for r in rows if "search_kernel" in r["Kernel_Name"]]
            ids = sorted({int(r["Dispatch_Id"]) for r in sel})[-steps:]                # the timed launches
            if pr.returncode != 0 or len(ids) < steps:
                log(f"[bench] live PMC pass {tag} FAILED: rc {pr.returncode}, {len(ids)} launches of the search kernel seen; child stdout tail: {so[-300:]!r}")
                return {"bytes": None, "note": f"PMC pass {tag} failed (rc {pr.returncode}, {len(ids)} launches of the search kernel seen)"}
            for c in counters:
                got[c] = sum(float(r["Counter_Value"]) for r in sel if int(r["Dispatch_Id"]) in ids and r["Counter_Name"] == c) / steps
            log(f"[bench] live PMC pass {tag} ({time.time() - t0:.0f}s): " + ", ".join(f"{c} = {got[c]:.4g}" for c in counters) + " per launch of the search kernel")
        except Exception as e:                                   # (a profiler problem must not cost the bench line)
            return {"bytes": None, "note": f"PMC pass {tag} raised {type(e).__name__}: {e}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd = int(got["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32)
    wr = int(got["WRITE_SIZE"] * 1024)
    io = int(got["TCC_EA0_RDREQ_IO_32B_sum"] * 32)
    return {"bytes": rd + wr, "hbm_read": rd, "hbm_write": wr, "pcie_read": io, "fetch_size_raw": int(got["FETCH_SIZE"] * 1024),
            "write_requests": int(got["TCC_EA0_WRREQ_sum"]),
            "note": f"HBM bytes per launch measured in THIS run (rocprofv3 --pmc, one pass per counter group, the {steps} timed launches of the same "
                    f"command): reads {rd / 1e9:.3f} GB = TCC_EA0_RDREQ_DRAM_32B x 32 (byte-exact on known byte counts: profiles/r04_traffic_calibration.md; "
                    f"FETCH_SIZE tallies every 128-byte request at 64: raw {got['FETCH_SIZE'] * 1024 / 1e9:.3f} GB) + writes {wr / 1e9:.3f} GB = WRITE_SIZE "
                    f"(32 B per scattered 4-byte store); {io / 1e9:.3f} GB more were read over PCIe (pulled adjacency rows)"}


# ---------------------------------------------------------------------------------------------------------- build first
def build_everything(rank, world):
    """libbang.so + bang_search, the oracle (checker / cpu_baseline leg) and the shape-workload generator, compiled by rank 0 before
    anything touches the GPU (compiling does not).  A child of live_traffic() (BANG_BENCH_NO_TRAFFIC=1) runs under rocprofv3 --pmc and
    must not compile anything: it raises on a stale or missing library instead."""
    import bang_amd
    from bang_amd import binding
    from oracle import oracle as O
    from tools import shape_workload
    if os.environ.get("BANG_BENCH_NO_TRAFFIC"):
        os.environ["BANG_NO_BUILD"] = "1"            # binding.build(), oracle.build(), shape_workload._lib(): check, never compile
    stamp = os.path.join(ROOT, "gpurun_out", f".bench_built_{os.environ.get('MASTER_PORT', '0')}")
    if rank == 0:
        bang_amd.build()
        O.build()
        shape_workload._lib()
        if world > 1:
            os.makedirs(os.path.dirname(stamp), exist_ok=True)
            open(stamp, "w").write(str(time.time()))
    elif world > 1:
        t0 = time.time()
        while not (os.path.exists(stamp) and os.path.getmtime(stamp) >= T_PROCESS_START - 5) and time.time() - t0 < 600:
            time.sleep(0.2)
    _ = binding


# ---------------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sift1b_shape", choices=sorted(WORKLOADS) + ["sift1b_shape", "deep100m_shape"],
                    help="default: sift1b_shape = BASELINE.json configs[3] / [4], the configuration the target number is quoted on")
    ap.add_argument("--shape-n", type=int, default=0, help="override N of a *_shape workload")
    ap.add_argument("--graph", default="", choices=["", "host", "device", "auto"],
                    help="host: graph in host RAM (BANG_Base placement; rows pulled by the kernel, or --pull 0: C++ walker); device: graph in HBM")
    ap.add_argument("--L", type=int, default=0, help="worklist length; 0 = smallest L on the harness grid with recall >= target (shape-only: 152)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--queries", type=int, default=0, help="override the batch size of the workload")
    ap.add_argument("--recall-target", type=float, default=90.0)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0, help="walker threads per lane; 0 = engine default")
    ap.add_argument("--batches", type=int, default=1, help="N > 1 throughput mode: every rank streams this many WHOLE batches per step")
    ap.add_argument("--pull", type=int, default=-1, choices=[-1, 0, 1],
                    help="host-graph placement: 1 = the kernel pulls adjacency rows over PCIe, 0 = C++ walker threads, -1 = engine default")
    ap.add_argument("--resident-graph", action="store_true",
                    help="sift1b_shape: build the whole 388-byte-per-node graph image in host memory (N then fits graph + pull rows) "
                         "instead of streaming the generator through the engine (default: streamed, N fits the pull rows)")
    ap.add_argument("--host-codes", action="store_true",
                    help="shape-only workloads: generate the PQ codes in HOST memory too (resident graph image, N must fit) so that the "
                         "oracle can check the results -- parity runs of the sharded job at reduced N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the side measurements (other configs, K2 alone)")
    ap.add_argument("--legs", default="", help="comma list of legs to run (default: k2,sift1m,deep100m,walker,sift300m,sift10m; on request: sift100m, sift1b)")
    ap.add_argument("--leg-budget-s", type=float, default=600.0,
                    help="a leg is skipped (and listed in config.legs_skipped) once the run -- counted from the start of the process -- has "
                         "taken this long: the default run stays within minutes")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for dry runs of the N>1 logic)")
    ap.add_argument("--no-events", action="store_true", help="do not stamp the launches of the timed steps")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC passes (profiles/) instead of two rocprofv3 passes of this run")
    args = ap.parse_args()

    ctx = Ctx()
    ctx.rank = rank = int(os.environ.get("RANK", "0"))
    ctx.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ctx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.k = k = args.k
    ctx.weak = False
    ctx.live_traffic = None
    # Everything that compiles is built HERE, before anything is profiled or touches the GPU: a child under `rocprofv3 --pmc` must never
    # start make / hipcc / gcc (every hop of such a tree would be an exec from a GPU-initialised process).  The children assert this.
    build_everything(rank, world)
    # A bench that is ITSELF being profiled (BANG_NO_BUILD is what every profiling script exports; the profiler's preloaded tool library
    # shows in LD_PRELOAD / ROCP_TOOL_LIBRARIES / ROCPROFILER_*) never starts the nested rocprofv3 passes: that launcher would be an exec from
    # a process whose GPU the preloaded library has already initialised.
    profiled = bool(os.environ.get("BANG_NO_BUILD") or os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprof" in os.environ.get("LD_PRELOAD", "")
                    or any(k.startswith("ROCPROF") for k in os.environ))
    if (world == 1 and not args.no_legs and not args.no_live_traffic and args.batches == 1 and not os.environ.get("BANG_BENCH_NO_TRAFFIC")
            and not profiled and not os.environ.get("BANG_BENCH_FORCE_GATHER")):
        # (before anything here touches the GPU: the profiled children need the HBM and must have exited by then)
        ctx.live_traffic = live_traffic(args, lambda *a: print(*a, file=sys.stderr, flush=True))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (libbang has no CPU fallback)")
    if os.environ.get("BANG_BENCH_SHARE_GPU"):          # dry run of the N>1 logic on a 1-GPU box: every rank on device 0
        ctx.local_rank = 0
    torch.cuda.set_device(ctx.local_rank)
    ctx.dev = dev = torch.device("cuda", ctx.local_rank)
    ctx.cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where collective buffers live
    # BANG_BENCH_FORCE_GATHER=1: run the N > 1 data path (bang_query_dev_e + the RCCL all-gather from device buffers) with ONE rank --
    # the only way to execute the RCCL call on a 1-GPU box (two ranks on one device are refused)
    ctx.force_gather = bool(os.environ.get("BANG_BENCH_FORCE_GATHER")) and world == 1
    if world > 1 or ctx.force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import bang_amd
    from oracle import oracle as O           # checker + cpu_baseline leg only
    if world > 1:
        dist.barrier()                       # (rank 0 built everything in build_everything(); the others waited for the files)

    lanes = args.lanes or int(os.environ.get("BANG_LANES", "0"))
    threads = args.threads or int(os.environ.get("BANG_THREADS", "0"))
    if world > 1 and not threads and args.pull == 0:
        # walker form: all ranks of the node share one CPU quota -- size the walker team from this rank's share of it
        share = max(1, usable_cpus() // world)
        threads = max(1, min(12, share - 1)) if os.environ.get("BANG_PERSISTENT", "-1") != "0" else max(1, min(4, share // max(1, lanes or 1)))

    # ------------------------------------------------------------------ the primary configuration (value / config / roofline)
    weak = world > 1 and args.batches > 1
    ctx.live_primary = ctx.live_traffic is not None
    prim = run_config(args.workload, ctx, args, O, graph=args.graph, pull=args.pull, L=args.L, steps=args.steps, warmup=args.warmup,
                      stream=(args.pull != 0 and not args.resident_graph), reserve_rows=(args.pull != 0), Q=args.queries,
                      shape_n=args.shape_n, batches=args.batches, lanes=lanes, threads=threads, keep=True, host_codes=args.host_codes)
    res, agg, L, graph = prim["res"], prim["res"]["agg"], prim["L"], prim["graph"]
    ix = prim["wl"]["ix"]
    ids_primary = res["ids"]
    m_primary, D_primary, dtype_primary = ix.m, ix.D, ix.dtype
    stride_primary = int(agg["code_stride"]) or ix.m           # the code-row layout the engine searched (K2 alone is measured on the same)
    out, cfg = None, {}
    if rank == 0:
        recall = prim["recall"]
        # the first 24 keys are what the driver's record keeps: the workload, the checks and the headline numbers of every leg come first
        # (filled in as the legs run -- a key keeps the position it was created at), the detail behind them
        cfg = {"workload": prim["wl"]["name"], "L": L, "k": k, "recall_at_10": (round(recall, 3) if recall == recall else None),
               "graph": graph,
               "parity_vs_oracle_first_64" if prim["structured"] else "result_properties_ok": prim["ok"],
               "reduced_n_hip_ids_equal_oracle": None, "k2_alone_frac": None, "k2_alone_GBps": None,
               "traffic_over_algorithmic": (res["roofline"] or {}).get("traffic_over_algorithmic"),
               "qps_incl_init": res["qps_incl_init"],
               "sift300m_qps": None, "sift300m_recall": None, "sift300m_L": None, "sift300m_parity_ok": None,
               "sift300m_hops_p50": None, "sift1m_qps": None, "sift1m_recall": None, "sift1m_parity_ok": None,
               "deep100m_shape_qps": None, "deep100m_shape_frac": None, "walker_qps": None,
               "adjacency_rows_also_in_hbm": agg["rows_in_hbm"], "legs_skipped": None,
               "graph_placement": prim["placement_note"] or f"{graph} (requested)",
               "lanes": agg["lanes"], "walker_threads": agg["walker_threads"],
               "search_kernel_workgroups": agg["workgroups"], "queries_per_workgroup": agg["wg_queries"],
               "iterations": agg["iterations"], "hops_p50": agg["hops_p50"], "hops_p99": agg["hops_p99"], "hops_max": agg["hops_max"],
               "host_loop": host_loop_name(agg, graph),
               "rerank_vectors": ("graph entries in HBM" if graph == "device" else
                                  "packed copy in HBM" if agg["vectors_on_device"] else "shipped by the walker (PCIe)"),
               "vector_dtype": ix.dtype, "pq_code_row_stride_bytes": agg["code_stride"], "batches_per_step": args.batches if weak else 1,
               "pcie_h2d_bytes_per_step": int(agg["h2d_bytes"] // args.steps),
               "pcie_pulled_bytes_per_step": int(agg["pulled_bytes"] // args.steps),
               "front_ms_per_step": round(agg["front_ms"] / args.steps, 3),
               "front_busy_ms_per_step": round(agg["front_busy_ms"] / args.steps, 3),
               "walker_ms_per_step": round(agg["walker_ms"] / args.steps, 3),
               "dist_evals_per_step": agg["dist_evals"] // args.steps,
               "filter_probes_per_step": 2 * agg["fetched"] // args.steps,
               "filter_loads_skipped_per_step": agg["filter_loads_skipped"] // args.steps,
               "gathered_ids_equal_oracle_whole_batch": prim["gathered_ok"],
               "search_ms_per_step_max_over_ranks": res.get("search_ms"), "gather_ms_per_step_max_over_ranks": res.get("gather_ms"),
               "step_ms_min": min(res["step_ms"]), "step_ms_max": max(res["step_ms"]), "step_ms": res["step_ms"]}
        out = {"metric": "queries/sec @ recall@10>=0.9, 10K-query batch", "value": res["queries_per_s"], "unit": "queries/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": ARITH_DTYPE,
               "data": "synthetic", "config": cfg, "roofline": res["roofline"], "cpu_baseline": None}

    want = set(x for x in args.legs.split(",") if x)
    legs = world == 1 and not args.no_legs and not os.environ.get("BANG_BENCH_NO_LEGS")

    t_run0 = T_PROCESS_START                 # (the budget covers the whole run: PMC passes + primary + legs)
    skipped = []

    def leg_on(name):
        if not (legs and (not want or name in want)):
            return False
        if time.time() - t_run0 > args.leg_budget_s:
            skipped.append(name)
            return False
        return True

    # the CPU baseline of a structured primary needs its index: before the release
    if rank == 0 and world == 1 and not args.no_cpu_baseline and prim["structured"]:
        out["cpu_baseline"] = cpu_baseline_structured(prim, O, k, prim["wl"]["queries"])
    if leg_on("L200") and prim["structured"] and L != 200:
        # BASELINE.json configs[1] also names L = 200: the same batch at that worklist length
        e = prim["eng"]
        e.free(); e.set_searchparams(k, 200); e.alloc(prim["Qr"])
        r2 = measure(e, prim["wl"], prim["my_q"], 200, 3, 1, ctx, graph)
        ids_o, _ = prim["orc"].search(prim["my_q"][:64], k, 200)
        cfg["at_L200"] = leg_summary(r2, prim["wl"], graph, recall=O.recall(prim["wl"]["gt_i"], prim["wl"]["gt_d"], r2["ids"], k),
                                     extra={"parity_vs_oracle_first_64": bool(np.array_equal(r2["ids"][:64], ids_o))})
        flat(cfg, "L200", cfg["at_L200"])
    release_config(prim)

    # ------------------------------------------------------------------ K2 alone (the stage the BASELINE metric quotes an HBM figure for)
    k2 = {}
    if leg_on("k2") and out is not None and out["roofline"] is not None:
        try:
            k2[f"m{m_primary}"] = k2_alone(D_primary, m_primary, dtype_primary, ctx, reps=10, stride=(stride_primary if stride_primary != m_primary else 0))
            out["roofline"]["k2_alone"] = k2[f"m{m_primary}"]
            cfg["k2_alone_frac"] = out["roofline"]["k2_alone_frac"] = k2[f"m{m_primary}"]["frac"]
            cfg["k2_alone_GBps"] = out["roofline"]["k2_alone_GBps"] = k2[f"m{m_primary}"]["achieved"]
        except Exception as ex:
            out["roofline"]["k2_alone"] = {"error": repr(ex)[:300]}

    # ------------------------------------------------------------------ CPU baseline beside a shape-only primary
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not prim["structured"]:
        try:
            out["cpu_baseline"] = cpu_baseline_shape(args.workload, ctx, args, O, L)
            cfg["reduced_n_hip_ids_equal_oracle"] = out["cpu_baseline"]["hip_ids_equal_oracle_on_sample"]
        except Exception as ex:
            out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": usable_cpus(), "kind": "port", "sample": "FAILED: " + repr(ex)[:300]}

    # ------------------------------------------------------------------ legs (single GPU only): the other BASELINE configurations
    leg_steps, leg_warm = 5, 1

    def guarded(key, fn):
        try:
            fn()
        except Exception as ex:                              # a leg must never take the primary line down
            import traceback
            log(f"[bench] leg {key} FAILED:\n" + traceback.format_exc())
            cfg[key] = {"error": repr(ex)[:300]}
            cfg[key.replace("at_", "") + "_error"] = repr(ex)[:120]
            torch.cuda.empty_cache()

    if leg_on("sift1m") and args.workload != "sift1m":
        def leg_sift1m():
            # configs[1]: SIFT1M-like, recall-verified; engine default placement (HBM), then the host placement in both loop forms
            r1 = run_config("sift1m", ctx, args, O, steps=leg_steps, warmup=leg_warm, keep=True)
            wl1 = r1["wl"]
            cfg["at_sift1m"] = leg_summary(r1["res"], wl1, r1["graph"], recall=r1["recall"], extra={"parity_vs_oracle_first_64": r1["ok"]})
            flat(cfg, "sift1m", cfg["at_sift1m"])
            ids1, L1, q1_ = r1["res"]["ids"], r1["L"], r1["my_q"]
            e = r1["eng"]
            e.free(); e.set_searchparams(k, 200); e.alloc(r1["Qr"])
            r2 = measure(e, wl1, q1_, 200, 3, 1, ctx, r1["graph"])
            ids_o, _ = r1["orc"].search(q1_[:64], k, 200)
            cfg["at_sift1m_L200"] = leg_summary(r2, wl1, r1["graph"], recall=O.recall(wl1["gt_i"], wl1["gt_d"], r2["ids"], k),
                                                extra={"parity_vs_oracle_first_64": bool(np.array_equal(r2["ids"][:64], ids_o))})
            flat(cfg, "sift1m_L200", cfg["at_sift1m_L200"])
            if not args.no_cpu_baseline:
                cb = cpu_baseline_structured(r1, O, k, wl1["queries"])
                cfg["at_sift1m"]["cpu_baseline"] = cb
                cfg["sift1m_cpu_qps"] = cb["value"]
            k2[f"m{wl1['ix'].m}"] = k2_alone(wl1["ix"].D, wl1["ix"].m, wl1["ix"].dtype, ctx)
            e.free(); e.unload(); e.close()
            r1["eng"] = None
            for key, pull in (("sift1m_host_pull", -1), ("sift1m_host_walker", 0)):
                e2 = make_engine(wl1, "host", ctx, timing=0 if args.no_events else 1, pull=pull)
                e2.set_searchparams(k, L1)
                e2.alloc(q1_.shape[0])
                r3 = measure(e2, wl1, q1_, L1, leg_steps, leg_warm, ctx, "host", traffic_key=("sift1m_host" if pull else None))
                cfg["at_" + key] = leg_summary(r3, wl1, "host", recall=O.recall(wl1["gt_i"], wl1["gt_d"], r3["ids"], k),
                                               extra={"ids_equal_device_run": bool(np.array_equal(r3["ids"], ids1))})
                flat(cfg, key, cfg["at_" + key])
                cfg[key + "_ids_equal_device_run"] = cfg["at_" + key]["ids_equal_device_run"]
                e2.free(); e2.unload(); e2.close()
            release_config(r1)
        guarded("at_sift1m", leg_sift1m)

    if leg_on("deep100m") and args.workload != "deep100m_shape":
        def leg_deep():
            r = run_config("deep100m_shape", ctx, args, O, steps=leg_steps, warmup=leg_warm, keep=True)
            cfg["at_deep100m_shape"] = leg_summary(r["res"], r["wl"], r["graph"], props=r["ok"])
            flat(cfg, "deep100m_shape", cfg["at_deep100m_shape"])
            ixd = r["wl"]["ix"]
            mm, DD, dt = ixd.m, ixd.D, ixd.dtype
            cs_d = int(r["res"]["agg"]["code_stride"]) or mm
            release_config(r)
            k2[f"m{mm}"] = k2_alone(DD, mm, dt, ctx, stride=(cs_d if cs_d != mm else 0))
        guarded("at_deep100m_shape", leg_deep)

    if leg_on("sift1b") and args.workload != "sift1b_shape":
        def leg_1b():
            r = run_config("sift1b_shape", ctx, args, O, steps=leg_steps, warmup=leg_warm, keep=True)
            cfg["at_sift1b_shape"] = leg_summary(r["res"], r["wl"], r["graph"], props=r["ok"])
            flat(cfg, "sift1b_shape", cfg["at_sift1b_shape"])
            release_config(r)
        guarded("at_sift1b_shape", leg_1b)

    if leg_on("walker"):
        def leg_walker():
            # the north-star data flow on its own configuration: a resident graph image served by the C++ walker threads
            r = run_config("sift1b_shape", ctx, args, O, graph="host", pull=0, steps=leg_steps, warmup=leg_warm, stream=False,
                           reserve_rows=False, traffic=False, keep=True)
            cfg["at_sift1b_shape_walker"] = leg_summary(r["res"], r["wl"], "host", props=r["ok"])
            flat(cfg, "walker", cfg["at_sift1b_shape_walker"])
            rf = r["res"]["roofline"] or {}
            if "pcie_h2d" in rf:
                cfg["walker_bar_GBps"] = rf["pcie_h2d"]["achieved_GBps"]
                cfg["at_sift1b_shape_walker"]["pcie_h2d"] = rf["pcie_h2d"]
            a = r["res"]["agg"]
            cfg["walker_leg_threads"] = a["walker_threads"]
            cfg["walker_step_ms_min"], cfg["walker_step_ms_max"] = min(r["res"]["step_ms"]), max(r["res"]["step_ms"])
            release_config(r)
        guarded("at_sift1b_shape_walker", leg_walker)

    if leg_on("sift300m"):
        def leg_300m():
            # SIFT1B's PQ layout and placement (graph in host RAM, rows pulled over PCIe) on a structured, recall-verified index three tenths of the
            # headline's size: "QPS @ recall >= 0.9" at 3e8 points, the first 64 queries against the oracle
            r = run_config("sift300m", ctx, args, O, graph="host", steps=leg_steps, warmup=leg_warm, traffic=False, keep=True)
            cfg["at_sift300m"] = leg_summary(r["res"], r["wl"], "host", recall=r["recall"], extra={"parity_vs_oracle_first_64": r["ok"]})
            flat(cfg, "sift300m", cfg["at_sift300m"])
            cfg["sift300m_hops_max"] = cfg["at_sift300m"]["hops_p50_p99_max"][2]
            release_config(r)
        guarded("at_sift300m", leg_300m)

    if "sift100m" in want and leg_on("sift100m"):
        def leg_100m():
            # SIFT1B's PQ layout and placement (graph in host RAM, rows pulled over PCIe) on a structured, recall-verified 1e8-point index
            r = run_config("sift100m", ctx, args, O, graph="host", steps=leg_steps, warmup=leg_warm, traffic=False, keep=True)
            cfg["at_sift100m"] = leg_summary(r["res"], r["wl"], "host", recall=r["recall"], extra={"parity_vs_oracle_first_64": r["ok"]})
            flat(cfg, "sift100m", cfg["at_sift100m"])
            cfg["sift100m_hops_max"] = cfg["at_sift100m"]["hops_p50_p99_max"][2]
            release_config(r)
        guarded("at_sift100m", leg_100m)

    if leg_on("sift10m"):
        def leg_10m():
            # a structured index beyond the Infinity Cache (N = 10 M: 320 MB of codes, 3.9 GB graph), recall-verified, both placements
            r = run_config("sift10m", ctx, args, O, graph="host", steps=leg_steps, warmup=leg_warm, keep=True)
            w3 = r["wl"]
            cfg["at_sift10m"] = leg_summary(r["res"], w3, "host", recall=r["recall"], extra={"parity_vs_oracle_first_64": r["ok"]})
            flat(cfg, "sift10m", cfg["at_sift10m"])
            e = r["eng"]
            e.free(); e.unload(); e.close()
            r["eng"] = None
            e4 = make_engine(w3, "device", ctx, timing=0 if args.no_events else 1)
            e4.set_searchparams(k, r["L"])
            e4.alloc(r["Qr"])
            r5 = measure(e4, w3, r["my_q"], r["L"], leg_steps, leg_warm, ctx, "device")
            cfg["at_sift10m_device_graph"] = leg_summary(r5, w3, "device", recall=O.recall(w3["gt_i"], w3["gt_d"], r5["ids"], k),
                                                         extra={"ids_equal_host_run": bool(np.array_equal(r5["ids"], r["res"]["ids"]))})
            flat(cfg, "sift10m_device", cfg["at_sift10m_device_graph"])
            e4.free(); e4.unload(); e4.close()
            release_config(r)
        guarded("at_sift10m", leg_10m)

    if skipped:
        cfg["legs_skipped"] = ",".join(skipped)
    if out is not None and out["roofline"] is not None and k2:
        out["roofline"]["k2_alone_other_layouts"] = {kk: v for kk, v in k2.items() if kk != f"m{m_primary}"}
        for kk, v in k2.items():
            if v:
                cfg[f"k2_alone_{kk}_frac"] = v["frac"]
    if world > 1 or ctx.force_gather:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
