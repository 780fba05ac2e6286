#!/usr/bin/env python3
"""Stand-alone measurement of the PQ-distance stage (K2, `bang_k_pqdist`) and of a plain random-row gather, to put
the fused front kernel's numbers in context.

For each PQ layout of the BASELINE configs (m = 32 / 70 / 74) a shape-only code table far larger than the caches is
built, every one of Q = 10 000 queries gets 64 random neighbour ids, and the distance kernel alone is timed with its
in-kernel s_memrealtime stamps.  Algorithmic bytes = evaluations x (m + 8)  (SURVEY 8(d)).

    python tools/k2_alone.py            # on the GPU box
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))

import bang_amd  # noqa: E402
from bang_amd import binding as B  # noqa: E402
from bang_amd.formats import Index, pack_graph  # noqa: E402
from bang_amd.synth import chunk_offsets  # noqa: E402


def run(N, D, m, dtype, Q=10_000, reps=20):
    rng = np.random.default_rng(1)
    npd = np.float32 if dtype == "float" else np.uint8
    codes = rng.integers(0, 256, (N, m), dtype=np.uint8)
    # minimal graph: the stage only needs codes + pivots; give every node a 1-entry adjacency so the Index is valid
    vec = np.zeros((2, D), npd)
    graph = pack_graph(vec, np.ones(2, np.uint32), np.zeros((2, 64), np.uint32))
    ix = Index(dtype, N, D, 64, m, 0, graph, codes, (rng.standard_normal((256, D)) * 30).astype(np.float32),
               np.zeros(D, np.float32), chunk_offsets(D, m))
    ix.N = N
    q = rng.integers(0, 255, (Q, D)).astype(npd)
    st = B.IterState.__new__(B.IterState)
    # IterState.__init__ reads the medoid's adjacency from ix.graph: build it by hand with a tiny graph but the big code table
    small = Index(dtype, 2, D, 64, m, 0, graph, codes[:2], ix.pivots, ix.centroid, ix.chunk_off)
    st.__init__(small, q, 16)
    st.d_codes = B.DeviceBuffer.from_numpy(codes, slack=256)
    lists = rng.integers(0, N, (Q, 64)).astype(np.uint32)
    cnt = np.full(Q, 64, np.uint32)
    nb = np.zeros((Q, B.NBR_STRIDE), np.uint32)
    nb[:, :64] = lists
    st.d_nbrs.upload(nb)
    st.d_cnt.upload(cnt)
    st.first, st.iter = 0, 2
    kt = B.DeviceBuffer(256 * 16 * (reps + 2))
    out = []
    for r in range(reps + 2):
        p = st.params()
        p.d_ktime = kt.ptr + r * 256 * 16
        B._check(B.lib().bang_k_pqdist(C.byref(p), None), "bang_k_pqdist")
    B.sync()
    t = kt.download(np.uint64, (reps + 2, 256, 2))
    for r in range(2, reps + 2):
        used = t[r][t[r, :, 0] > 0]
        out.append((used[:, 1].max() - used[:, 0].min()) * 1e-2)      # us (100 MHz ticks)
    us = float(np.median(out))
    evals = Q * 64
    return {"m": m, "D": D, "dtype": dtype, "N": N, "psz_mp": [st.psz, st.mp], "evals_per_launch": evals,
            "median_us": round(us, 2), "algorithmic_GBps": round(evals * (m + 8) / us / 1e3, 1),
            "code_bytes_GBps": round(evals * m / us / 1e3, 1)}


def big():
    """>= 1 ms per launch on a 4 GB code table (far beyond the 256 MB Infinity Cache): what bench.py reports as roofline.k2_alone."""
    import torch
    sys.path.insert(0, ROOT)
    import bench

    class Ctx:
        pass
    ctx = Ctx()
    ctx.dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = []
    for D, m, dt, stride in ((128, 32, "uint8", 0), (128, 70, "uint8", 0), (128, 70, "uint8", 128), (96, 74, "float", 0), (96, 74, "float", 128)):
        r = bench.k2_alone(D, m, dt, ctx, reps=10, stride=stride)
        out.append(r)
        print(json.dumps(r), flush=True)
    return out


if __name__ == "__main__":
    bang_amd.build()
    if "--big" in sys.argv:
        big()
        sys.exit(0)
    if os.environ.get("K2_SIZE_SWEEP"):      # table size vs the 32 MB of L2 and the 256 MB Infinity Cache (m = 32: 32 B rows)
        res = [run(n, 128, 32, "uint8") for n in (1_000_000, 3_000_000, 6_000_000, 12_000_000, 40_000_000)]
    else:
        res = [run(40_000_000, 128, 32, "uint8"), run(20_000_000, 128, 70, "uint8"), run(20_000_000, 96, 74, "float")]
    for r in res:
        print(json.dumps(r))
