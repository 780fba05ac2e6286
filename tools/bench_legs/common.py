"""bench.py's shared machinery: workloads, engines, one timed measurement (`measure`), one configuration (`run_config`).
bench.py re-exports everything here, so `import bench; bench.measure(...)` keeps working for the tools."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T_PROCESS_START = time.time()
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
            ctx.batch_ids = dg.batch_ids(copy=False)               # the batch's answer reaches the host on one rank (one D2H copy)
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
                "workgroups", "hops_p50", "hops_p99", "hops_max", "graph_pull", "code_stride", "rows_in_hbm", "rerank_fused", "walker_rows")
    agg = dict(front_ms=0.0, front_busy_ms=0.0, walker_ms=0.0, sync_ms=0.0, enqueue_ms=0.0, dist_evals=0, front_launches=0,
               fetched=0, candidates=0, h2d_bytes=0, pulled_bytes=0, filter_loads_skipped=0, rows_from_peer=0, rows_from_own_hbm=0)
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
    if ctx.world > 1:                                   # what every rank saw itself (means over the steps): bang_query of its shard, the collective
        mine = torch.tensor([1e3 * sum(search_s) / steps, 1e3 * sum(gather_s) / steps], dtype=torch.float64, device=ctx.cdev)
        every = [torch.zeros_like(mine) for _ in range(ctx.world)]
        dist.all_gather(every, mine)
        res["search_ms_per_rank"] = [round(float(t[0].item()), 4) for t in every]
        res["gather_ms_per_rank"] = [round(float(t[1].item()), 4) for t in every]
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
        elif live and not live.get("bytes"):
            traffic_note = "HBM traffic not measured in this run: " + str(live.get("note"))      # (why the live PMC passes gave nothing)
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
                "kernel": ("search_kernel: ONE launch per batch = every iteration of every query (K5 filter, K2 PQ distance, K4 parent, K3a sort, "
                           "K3b merge; K6+K7 re-rank where fused)") if persistent else "front_kernel (K5 filter + K2 PQ distance + K4 parent, fused)",
                "timer": "in-kernel s_memrealtime (100 MHz), every timed launch; rocprofv3 --kernel-trace agrees (profiles/)"}
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
    if a.get("walker_rows"):
        return "search kernel, host-paced: C++ walker threads read the 256-byte adjacency rows and write them through the PCIe BAR"
    return "search kernel, host-paced: C++ walker threads read graph entries and write the adjacency rows through the PCIe BAR"


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
    # N > 1, rows pulled: the node's adjacency rows go into the node's spare HBM -- rank r keeps slice r, every rank maps the others over IPC
    # (bang_amd/shard.py share_rows; BANG_BENCH_NO_PEER_ROWS=1: every rank pulls from host DRAM as in rounds 2-4)
    ctx.peer_rows = None
    if ctx.world > 1 and graph != "device" and pull != 0 and not os.environ.get("BANG_BENCH_NO_PEER_ROWS"):
        import torch.distributed as dist
        from bang_amd import shard as _shard
        try:                                                 # (share_rows agrees on failure across the ranks by itself: all raise, or none)
            ctx.peer_rows = _shard.share_rows(eng, ctx.rank, ctx.world, int(ix.N), slice_rows=int(os.environ.get("BANG_BENCH_PEER_SLICE_ROWS", "0")))
            log(f"[bench] peer rows: {ctx.peer_rows['slice_rows']} rows per rank, {100 * ctx.peer_rows['fraction']:.1f} % of the adjacency rows in the node's HBM")
        except Exception as ex:                              # noqa: BLE001  (no IPC here, not pull mode ...: host rows serve everything)
            log(f"[bench] peer rows not available: {ex!r}")
            ctx.peer_rows = {"error": repr(ex)[:200]}
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
               placement_note=placement_note, structured=gt_i is not None, name=name, gathered_ok=gathered_ok,
               peer_rows=bool(ctx.peer_rows and not ctx.peer_rows.get("error")))
    if not keep:
        release_config(out)
    return out


def release_config(rc):
    import torch
    if rc.get("eng") is not None:
        e = rc["eng"]
        e.free()
        if rc.get("peer_rows"):                              # (two phases: close the peers' mappings, meet, then free what they had mapped)
            from bang_amd import shard as _shard
            _shard.unshare_rows(e)
        e.unload(); e.close()
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


