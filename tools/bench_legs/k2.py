"""bench.py leg: K2 alone (the stage the BASELINE metric quotes an HBM figure for)."""
import ctypes as C

import numpy as np

from .common import HBM_PEAK_GBPS

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
    # the same launch under rocprofv3 --kernel-trace, from the committed record of tools/profile_k2.sh (the stage BASELINE.json quotes an HBM figure for)
    try:
        import json
        import os
        from .common import ROOT
        tr = json.load(open(os.path.join(ROOT, "profiles", "k2_alone_trace.json")))
        lay = tr["layouts"].get(f"m{m}_stride128" if rb == 128 and m != 128 else f"m{m}_packed")
        if lay:
            out["rocprof_kernel_trace_avg_us"] = lay["avg_us"]
            out["rocprof_record"] = "profiles/r06_k2_alone.md"
    except Exception:
        pass
    del codes, nbrs, dist_o, cnt, qc, packed
    torch.cuda.empty_cache()
    return out


