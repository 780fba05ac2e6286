#!/usr/bin/env python3
"""Builder experiments (GPU box): build a structured index with index_build.make_index_large under a few parameter sets and print
build time, recall per L on the harness grid and hops -- how good a graph the batched Vamana-style builder makes.
    python tools/dev/build_exp.py N [variant ...]      variants: r2 (round-2 one-pass), rev (reverse edges), rev8 ... see VARIANTS"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
import numpy as np  # noqa: E402

VARIANTS = {
    "r2": dict(reverse=False, n_random=32, cells=-1),          # round 2: one prune pass, R/2 random links, single-level partition
    "rev": dict(reverse=True, n_random=12),
    "rev8": dict(reverse=True, n_random=8),
    "rev16": dict(reverse=True, n_random=16),
    "rev12k64": dict(reverse=True, n_random=12, K=64, probes=8),
    "rev12a1": dict(reverse=True, n_random=12, alpha=1.0),
    "rev24": dict(reverse=True, n_random=24),
    "r2m70": dict(reverse=False, n_random=32, m=70),
    "revm70": dict(reverse=True, n_random=12, m=70),
    "r2m64": dict(reverse=False, n_random=32, m=64),
    "revm70p12": dict(reverse=True, n_random=12, m=70, probes=12),
    "swm70": dict(reverse=True, n_random=16, m=70, links="smallworld"),
    "swm70r24": dict(reverse=True, n_random=24, m=70, links="smallworld"),
    "unim70": dict(reverse=True, n_random=16, m=70, links="uniform"),
    "swgm": dict(reverse=True, n_random=16, m=70, select="groupmin"),
    "swgmp12": dict(reverse=True, n_random=16, m=70, select="groupmin", probes=12),
    "swp10": dict(reverse=True, n_random=16, m=70, probes=10),
    "gm6": dict(m=70, select="groupmin", probes=6),
    "gm8": dict(m=70, select="groupmin", probes=8),
    "gm6c1k": dict(m=70, select="groupmin", probes=8, cell=1024),
    "gm6k32": dict(m=70, select="groupmin", probes=6, K=32),
    "revm70n1": dict(reverse=True, n_random=12, m=70, noise=1.0),
    "revm70n05": dict(reverse=True, n_random=12, m=70, noise=0.5),
}


def main():
    import torch
    import bang_amd
    from bang_amd import index_build
    from oracle import oracle as O
    N = int(sys.argv[1])
    names = sys.argv[2:] or ["r2", "rev"]
    Q, k = 10000, 10
    for name in names:
        kw = dict(VARIANTS[name])
        m = kw.pop("m", 32)
        if kw.get("cells") == -1:
            kw["cells"] = int(max(16, min(8192, round((N ** 0.5) / 1.5))))
        t0 = time.time()
        ix, q, gt_i, gt_d = index_build.make_index_large(N, 128, "uint8", 64, m, Q, K=k, n_clusters=max(256, N // 10000), device="cuda",
                                                         log=lambda *a: print(*a, flush=True), diag=True, **kw)
        tb = time.time() - t0
        deg = ix.degrees()
        print(f"== {name} N={N}: built in {tb:.1f}s, degree mean {deg.mean():.1f} min {deg.min()}", flush=True)
        t0 = time.time()
        with bang_amd.Engine("uint8", graph=bang_amd.GRAPH_HOST) as e:
            e.load_index(ix)
            print(f"   engine load (host placement, pull rows) {time.time() - t0:.1f}s", flush=True)
            for L in range(10, 200, 12):
                e.set_searchparams(k, L)
                e.alloc(Q)
                e.init(Q)
                t1 = time.perf_counter()
                ids, _ = e.query(q)
                dt = time.perf_counter() - t1
                st = e.stats()
                e.free()
                r = O.recall(gt_i, gt_d, ids, k)
                print(f"   L={L:3d} recall={r:6.2f}  {Q / dt / 1e6:.2f} MQPS  hops p50 {st['hops_p50']} evals/query {st['dist_evals'] // Q}", flush=True)
                if r >= 97.0:
                    break
            e.unload()
        del ix
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
