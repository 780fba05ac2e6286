"""bench.py leg: the CPU baseline -- the oracle (C + OpenMP) on the GPU box's host cores, beside the HIP engine on the same index."""
import os
import time

import numpy as np

from .common import log, make_engine, run_once, usable_cpus

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


def cpu_baseline_shape(name, ctx, args, O, L, n=20_000_000, budget_s=10.0, max_reps=12):
    """The oracle beside a shape-only workload: the same layout (dtype, D, R, m, L, iteration cap, ~56 evaluations per
    iteration) on an index of REDUCED N whose PQ codes also exist in host memory (the full-size index keeps them only in HBM).
    n = 2e8 (graph entries 78 GB + codes 14 GB: far beyond every cache and most of the TLB reach) is the figure reported as
    cpu_baseline.value; n = 2e7 rides along as small_n_value.  The HIP engine runs the SAME reduced index first: its ids must equal
    the oracle's on every query, which puts an oracle-checked run of this layout into every bench run."""
    from tools import shape_workload
    k = ctx.k
    t0 = time.time()
    ix, queries, _, _, _, wl_name, _ = shape_workload.make(name, ctx.dev, n_override=n, Q=10_000, log=log, host_codes=True, planned=True)
    wl = dict(ix=ix, queries=queries, gt_i=None, gt_d=None, d_codes=None, name=wl_name, graph="host", prefix=None, shared_dir=None)
    eng = make_engine(wl, "host", ctx, timing=0)
    eng.set_searchparams(k, L)
    eng.alloc(queries.shape[0])
    ids_g = run_once(eng, queries, ctx)[0]
    eng.free(); eng.unload(); eng.close()
    nthreads = usable_cpus()
    orc = O.Oracle(ix)
    orc.search(queries[:256], k, L, nthreads=nthreads)               # warm
    # the whole 10 K batch, again and again until ~budget_s of CPU work are on the clock (the first pass also checks the HIP engine's ids)
    done, t_cpu, ids_o, reps = queries.shape[0], 0.0, None, 0
    while t_cpu < budget_s and reps < max_reps:
        t_a = time.perf_counter()
        ids_r, _ = orc.search(queries, k, L, nthreads=nthreads)
        t_cpu += time.perf_counter() - t_a
        reps += 1
        if ids_o is None:
            ids_o = ids_r
    parity = bool(np.array_equal(ids_g, ids_o))
    n_real = int(ix.N)
    shape_workload.release(ix)
    log(f"[bench] cpu baseline ({name}, N={n_real}): {reps} x {done} queries in {t_cpu:.1f}s on {nthreads} threads, parity with the HIP engine: {parity} "
        f"({time.time() - t0:.0f}s in all)")
    return {"value": round(done * reps / t_cpu, 1), "unit": "queries/s", "cores": nthreads, "kind": "port",
            "sample": f"{reps} x the {done}-query batch ({t_cpu:.0f} s of CPU work) of the {name} layout (m={ix.m}, L={L}, iteration cap L+49) at reduced N={n_real} with the "
                      f"PQ codes in host memory ({n_real * ix.entry_len / 1e9:.0f} GB of graph entries + {n_real * ix.m / 1e9:.0f} GB of codes), through oracle/ (C + OpenMP, "
                      f"{nthreads} threads = the CPU quota of this box; {os.cpu_count()} hardware threads visible), search only",
            "N": n_real, "hip_ids_equal_oracle_on_sample": parity}
