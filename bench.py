#!/usr/bin/env python3
"""bench.py -- queries/sec @ 10-recall@10 >= 0.9 on a 10K-query batch (BASELINE.json metric), the roofline of the search
kernel and of the PQ-distance stage (K2) alone, and the CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sift1b_shape|sift1m|sift10m|deep100m_shape|small|tiny]
                    [--graph host|device|auto] [--pull -1|0|1] [--batches B] [--no-legs] [--legs k2,sift1m,...]

One process per GPU.  `--gpus N` IS the rank count: under a launcher (torchrun sets RANK / LOCAL_RANK / WORLD_SIZE) WORLD_SIZE must equal N;
without one, N > 1 makes this process start its N ranks itself (tools/bench_legs/launch.py: fresh children before anything touches the GPU,
rank 0's line relayed, the children's status returned); fewer visible devices than N is an error, never a smaller run.

* For EVERY N the workload is `sift1b_shape` = BASELINE.json configs[3] (N = 1) / configs[4] (N > 1): the configuration the target
  number is quoted on (BANG_Base/test_driver.cpp:433-439, Cost_Analysis.pdf p.3) -- uint8, D = 128, R = 64, m = 70, 10 000 queries,
  L = 152, graph in host RAM.  The index is streamed through bang_load_stream_e (vectors + PQ codes into HBM, adjacency lists into
  256-byte pull rows in pinned host memory) and searched by ONE self-paced search-kernel launch per batch that pulls the rows over
  PCIe.  N is scaled to what the box's host memory holds (stated in config.workload).  `value`, `config`, `roofline` (that launch +
  `k2_alone` at m = 70) and `cpu_baseline` (the oracle on the same layout at reduced N, where the HIP engine is also checked
  against it) all belong to this configuration.
* N = 1 also carries LEGS (tools/bench_legs/), as nested objects `config.at_*` and -- because the driver's record keeps scalars only --
  flattened into `config.<leg>_qps`, `_recall`, `_L`, `_frac` ...  On the primary engine (ONE load of the 1e9-point index): `shards`
  (Q / 2, Q / 4, Q / 8 queries of the batch = one rank's shard of a 2 / 4 / 8-GPU job, the world-size-1 RCCL gather, and the projected
  speed-ups WITH the gather) and `walker` (the north-star data flow: C++ walker threads serve the host-paced kernel from the same
  256-byte rows).  On their own indices: `sift300m` (structured, recall-verified, SIFT1B's layout and placement: the recall-gated
  number, copied to config.recall_gated_*), `sift1m` (configs[1]; plus `sift1m_host_pull`, `sift1m_host_walker`, `sift1m_L200`),
  `deep100m_shape` (configs[2]), `sift10m`.
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
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))

import numpy as np  # noqa: E402

from tools.bench_legs import common as _common  # noqa: E402
from tools.bench_legs.common import *  # noqa: E402,F401,F403  (Ctx, build_workload, make_engine, measure, run_config ...: the tools import them from here)
from tools.bench_legs.common import (ARITH_DTYPE, T_PROCESS_START, WORKLOADS, Ctx, host_loop_name, log, release_config, run_config,  # noqa: E402
                                     usable_cpus)
from tools.bench_legs.cpu import cpu_baseline_shape, cpu_baseline_structured  # noqa: E402
from tools.bench_legs.k2 import k2_alone  # noqa: E402
from tools.bench_legs.traffic import live_traffic  # noqa: E402
from tools.bench_legs import ceiling as _ceiling, launch as _launch, legs as _legs, shards as _shards  # noqa: E402


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
        _ceiling.build()
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
    ap.add_argument("--legs", default="", help="comma list of legs to run (default: shards,walker,ceiling,k2,sift300m,sift1m,deep100m,sift10m; on request: sift100m, sift1b)")
    ap.add_argument("--leg-budget-s", type=float, default=600.0,
                    help="a leg is skipped (and listed in config.legs_skipped) once the run -- counted from the start of the process -- has "
                         "taken this long: the default run stays within minutes")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for dry runs of the N>1 logic)")
    ap.add_argument("--no-events", action="store_true", help="do not stamp the launches of the timed steps")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC passes (profiles/) instead of two rocprofv3 passes of this run")
    args = ap.parse_args()

    # --gpus N is the rank count, whoever starts the ranks (tools/bench_legs/launch.py).  Without a launcher and N > 1 this process starts the N
    # ranks itself -- before it or anything it imported has touched the GPU --, relays rank 0's line and exits with their status; with a launcher
    # a WORLD_SIZE that contradicts --gpus is an error; fewer visible devices than ranks is an error (never a silent N = 1 run).
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and not _launch.under_launcher():
        raise SystemExit(_launch.self_launch(args.gpus, sys.argv[1:], os.path.abspath(__file__)))
    _launch.check_world(args.gpus, int(os.environ.get("WORLD_SIZE", "1")))
    if args.gpus > 1:
        _launch.check_devices(args.gpus)

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
    world_seen = None
    if world > 1 or ctx.force_gather:                        # proof of the rank count in the record: an all-reduce of ones over the job's backend (every rank)
        ones = torch.ones(1, dtype=torch.int64, device=ctx.cdev)
        dist.all_reduce(ones)
        world_seen = int(ones.item())
    if rank == 0:
        recall = prim["recall"]
        # the first ~30 keys are what the driver's record keeps: first the number the metric's WORDING describes -- recall-gated, on a structured
        # index (the sift300m leg: filled in when it has run) --, then the workload, the checks, the shard sweep behind the multi-GPU projection
        # and the headline numbers of every leg (a key keeps the position it was created at), the detail behind them
        cfg = {"recall_gated_qps": None, "recall_gated_N": None, "recall_gated_recall": None, "recall_gated_L": None,
               "workload": prim["wl"]["name"], "L": L, "k": k, "recall_at_10": (round(recall, 3) if recall == recall else None),
               "graph": graph,
               "parity_vs_oracle_first_64" if prim["structured"] else "result_properties_ok": prim["ok"],
               "reduced_n_hip_ids_equal_oracle": None, "k2_alone_frac": None,
               "request_frac": None, "requests_G_per_s": None,
               "traffic_over_algorithmic": (res["roofline"] or {}).get("traffic_over_algorithmic"),
               "qps_incl_init": res["qps_incl_init"], "rerank_fused": int(agg.get("rerank_fused", 0)),
               "shard_ms_5000": None, "shard_ms_2500": None, "shard_ms_1250": None, "gather_ms_world1": None,
               "projected_speedup_2": None, "projected_speedup_4": None, "projected_speedup_8": None,
               "walker_qps": None, "sift1m_qps": None, "deep100m_shape_qps": None, "deep100m_shape_frac": None, "rccl_world_seen": None,
               "sift300m_qps": None, "sift1m_recall": None, "request_ceiling_G_per_s": None, "k2_alone_GBps": None, "walker_N": None,
               "sift300m_recall": None, "sift300m_L": None, "sift300m_parity_ok": None,
               "sift300m_hops_p50": None, "sift1m_parity_ok": None,
               "adjacency_rows_also_in_hbm": agg["rows_in_hbm"], "legs_skipped": None,
               "peer_rows": getattr(ctx, "peer_rows", None),
               "rows_from_peer_hbm_per_step": int(agg["rows_from_peer"] // args.steps), "rows_from_own_hbm_per_step": int(agg["rows_from_own_hbm"] // args.steps),
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
        cfg["rccl_world_seen"] = world_seen
        shape_only = not prim["structured"]
        # (the driver's record keeps ~120 characters of a string: the facts first)
        metric = (f"queries/sec, 10K-query batch; SHAPE-ONLY index, L={L}, {agg['iterations']} iterations/query (cap-bound), recall n/a; recall-gated: "
                  "config.recall_gated_* (structured index, 10-recall@10 >= 0.9). A shape-only batch is heavier per query than real data at recall 0.9"
                  if shape_only else "queries/sec @ 10-recall@10 >= 0.9, 10K-query batch")
        out = {"metric": metric, "value": res["queries_per_s"], "unit": "queries/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": ARITH_DTYPE,
               "data": "synthetic", "config": cfg, "roofline": res["roofline"], "cpu_baseline": None,
               # who started the ranks ("self": plain `python bench.py --gpus N`; "launcher": torchrun) and what the collective's backend counted
               "ranks_started_by": ("self" if os.environ.get("BANG_BENCH_SELF_LAUNCHED") else "launcher") if world > 1 else None,
               "world_seen": world_seen,
               # N > 1, rows pulled: True = the node's adjacency rows could NOT be shared over hipIpc and every rank pulls every row from host DRAM
               "peer_rows_fallback": bool(isinstance(getattr(ctx, "peer_rows", None), dict) and ctx.peer_rows.get("error")) if world > 1 else None,
               "search_ms_per_rank": res.get("search_ms_per_rank"), "gather_ms_per_rank": res.get("gather_ms_per_rank")}

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

    run = _legs.Run(ctx, args, O, cfg, {}, k)
    k2 = run.k2

    def guarded(key, fn):
        try:
            fn()
        except Exception as ex:                              # a leg must never take the primary line down
            import traceback
            log(f"[bench] leg {key} FAILED:\n" + traceback.format_exc())
            cfg[key] = {"error": repr(ex)[:300]}
            cfg[key.replace("at_", "") + "_error"] = repr(ex)[:120]
            torch.cuda.empty_cache()

    # the CPU baseline of a structured primary needs its index: before the release
    if rank == 0 and world == 1 and not args.no_cpu_baseline and prim["structured"]:
        out["cpu_baseline"] = cpu_baseline_structured(prim, O, k, prim["wl"]["queries"])
    if leg_on("L200") and prim["structured"] and L != 200:
        guarded("at_L200", lambda: _legs.leg_L200(run, prim, graph))
    # ------------------------------------------------------------------ legs on the PRIMARY engine (one load of the headline index)
    pulled = bool(agg.get("graph_pull")) and args.batches == 1
    if leg_on("shards") and pulled:
        guarded("at_shards", lambda: _shards.leg_shards(run, prim, L, res["ms_per_step"]))
    if leg_on("walker") and pulled:
        guarded("at_sift1b_shape_walker", lambda: _shards.leg_walker_rows(run, prim, L))
    release_config(prim)

    # ------------------------------------------------------------------ the request ceiling: what bounds a stream of random, never re-used requests past L2
    if leg_on("ceiling") and out is not None and out["roofline"] is not None:
        try:
            ceil = _ceiling.request_ceiling()
            rq = _ceiling.request_roofline(ctx.live_traffic, out["roofline"].get("avg_launch_us"), ceil)
            out["roofline"]["requests"] = rq if rq else {"ceiling": ceil, "note": "no live PMC passes in this run: the launch's own request count is not known"}
            cfg["request_ceiling_G_per_s"] = ceil.get("search_mix")
            if rq:
                cfg["requests_G_per_s"], cfg["request_frac"] = rq["achieved"], rq["frac"]
        except Exception as ex:
            out["roofline"]["requests"] = {"error": repr(ex)[:300]}

    # ------------------------------------------------------------------ K2 alone (the stage the BASELINE metric quotes an HBM figure for)
    if leg_on("k2") and out is not None and out["roofline"] is not None:
        try:
            k2[f"m{m_primary}"] = k2_alone(D_primary, m_primary, dtype_primary, ctx, reps=10, stride=(stride_primary if stride_primary != m_primary else 0))
            out["roofline"]["k2_alone"] = k2[f"m{m_primary}"]
            cfg["k2_alone_frac"] = out["roofline"]["k2_alone_frac"] = k2[f"m{m_primary}"]["frac"]
            cfg["k2_alone_GBps"] = out["roofline"]["k2_alone_GBps"] = k2[f"m{m_primary}"]["achieved"]
        except Exception as ex:
            out["roofline"]["k2_alone"] = {"error": repr(ex)[:300]}

    # ------------------------------------------------------------------ CPU baseline beside a shape-only primary: the oracle on the same layout at
    # N = 2e8 (78 GB of graph entries + 14 GB of codes: what a host-side search of this layout pays in cache and TLB misses) -- the reported value --
    # and at N = 2e7 (small_n_value: rounds 1-4 reported that one)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not prim["structured"]:
        try:
            small = cpu_baseline_shape(args.workload, ctx, args, O, L, n=int(os.environ.get("BANG_CPU_BASELINE_N", "20000000")))
            cb = small
            if not os.environ.get("BANG_CPU_BASELINE_SMALL_ONLY") and time.time() - t_run0 < args.leg_budget_s:
                try:
                    cb = cpu_baseline_shape(args.workload, ctx, args, O, L, n=int(os.environ.get("BANG_CPU_BASELINE_BIG_N", "200000000")), budget_s=12.0, max_reps=8)
                except Exception as ex:                      # (a box too small for the 2e8 index: the small one stands)
                    log(f"[bench] cpu baseline at the large N failed ({ex!r}): reporting the small-N figure")
            if cb is not small:
                cb["small_n_value"], cb["small_n"] = small["value"], small["N"]
                cb["hip_ids_equal_oracle_on_sample"] = bool(cb["hip_ids_equal_oracle_on_sample"] and small["hip_ids_equal_oracle_on_sample"])
            out["cpu_baseline"] = cb
            cfg["reduced_n_hip_ids_equal_oracle"] = cb["hip_ids_equal_oracle_on_sample"]
        except Exception as ex:
            out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": usable_cpus(), "kind": "port", "sample": "FAILED: " + repr(ex)[:300]}

    # ------------------------------------------------------------------ legs on their own indices (single GPU only): the other BASELINE configurations
    if leg_on("sift300m"):
        guarded("at_sift300m", lambda: _legs.leg_structured(run, "sift300m"))
        if cfg.get("sift300m_qps") is not None:              # the number the metric's wording describes: recall-gated, on a structured index
            cfg["recall_gated_qps"], cfg["recall_gated_N"] = cfg["sift300m_qps"], cfg.get("sift300m_N")
            cfg["recall_gated_recall"], cfg["recall_gated_L"] = cfg.get("sift300m_recall"), cfg.get("sift300m_L")
    if leg_on("sift1m") and args.workload != "sift1m":
        guarded("at_sift1m", lambda: _legs.leg_sift1m(run))
    if leg_on("deep100m") and args.workload != "deep100m_shape":
        guarded("at_deep100m_shape", lambda: _legs.leg_deep100m(run))
    if leg_on("sift1b") and args.workload != "sift1b_shape":
        guarded("at_sift1b_shape", lambda: _legs.leg_sift1b(run))
    if "sift100m" in want and leg_on("sift100m"):
        guarded("at_sift100m", lambda: _legs.leg_structured(run, "sift100m"))
    if leg_on("sift10m"):
        guarded("at_sift10m", lambda: _legs.leg_sift10m(run))

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

