#!/usr/bin/env python3
"""Which adjacency rows does a batch really read -- and what would a copy of n rows in HBM (engine option rows_hbm, pull mode) hit under
different choices of WHICH rows?  Runs one batch on a structured index, takes the candidate log (bang_get_candidate_log: the expanded
nodes of every query, in order) and scores

    first-n      the rows of the nodes [0, n): what cache_rows_in_hbm() copies (membership = one compare, no indirection)
    bfs          the first n nodes in breadth-first order from the medoid ("the rows every query reads first")
    in-degree    the n nodes most often named by other nodes' adjacency lists (static popularity)
    oracle-best  the n nodes this very batch expanded most often (the ceiling any static choice could reach on this batch)

as the share of row reads served from HBM, for n = 5 / 10 / 25 % of N, also split by hop number (where in a query's life the hits fall).

    python tools/rows_cache_study.py [--n 10000000] [--m 32] [--L 0]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
import numpy as np  # noqa: E402


def bfs_order(adj, deg, start, want):
    """node ids in breadth-first order from `start` (level by level, ascending id within a level), the first `want` of them"""
    N = adj.shape[0]
    seen = np.zeros(N, bool)
    seen[start] = True
    out, frontier, got = [np.array([start], np.int64)], np.array([start], np.int64), 1
    while got < want and frontier.size:
        rows = adj[frontier]
        ok = np.arange(adj.shape[1])[None, :] < deg[frontier][:, None]
        nxt = np.unique(rows[ok])
        nxt = nxt[~seen[nxt]]
        seen[nxt] = True
        out.append(nxt)
        got += nxt.size
        frontier = nxt
    return np.concatenate(out)[:want]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--m", type=int, default=32)
    ap.add_argument("--L", type=int, default=0)
    ap.add_argument("--queries", type=int, default=10_000)
    a = ap.parse_args()
    import bang_amd
    from bang_amd import index_build
    from oracle import oracle as O
    bang_amd.build()
    log = lambda *x: print(*x, file=sys.stderr, flush=True)   # noqa: E731
    kw = dict(select="groupmin", probes=12) if a.n > 20_000_000 else {}
    ix, q, gt_i, gt_d = index_build.make_index_large(a.n, 128, "uint8", 64, a.m, a.queries, K=10, n_clusters=max(256, a.n // 10000), device="cuda", log=log, **kw)
    N, Q = ix.N, q.shape[0]
    with bang_amd.Engine("uint8", graph=bang_amd.GRAPH_HOST) as e:
        e.load_index(ix)
        L = a.L
        if not L:
            for cand in range(10, 513, 12):
                e.set_searchparams(10, cand); e.alloc(Q); e.init(Q)
                ids, _ = e.query(q)
                e.free()
                if O.recall(gt_i, gt_d, ids, 10) >= 90.0:
                    L = cand
                    break
        e.set_searchparams(10, L); e.alloc(Q); e.init(Q)
        ids, _ = e.query(q)
        rec = O.recall(gt_i, gt_d, ids, 10)
        cand, cnt = e.candidate_log(Q, L)
        e.free(); e.unload()
    # row reads of the batch: every expanded node but the medoid at position 0 (its list is the seed, already on the device)
    hop = np.tile(np.arange(cand.shape[1])[None, :], (Q, 1))
    live = (hop >= 1) & (hop < cnt[:, None])
    reads, hops = cand[live].astype(np.int64), hop[live]
    adj, deg = ix.adjacency(), ix.degrees()
    indeg = np.bincount(adj[np.arange(adj.shape[1])[None, :] < deg[:, None]].ravel().astype(np.int64), minlength=N)
    freq = np.bincount(reads, minlength=N)
    out = {"N": N, "Q": Q, "L": L, "recall": round(rec, 2), "row_reads_per_batch": int(reads.size), "distinct_nodes_read": int((freq > 0).sum()),
           "hops_median": int(np.median(cnt)), "orders": {}}
    for share in (0.05, 0.10, 0.25):
        n = int(N * share)
        orders = {"first-n": np.arange(n), "bfs": bfs_order(adj, deg, int(ix.medoid), n), "in-degree": np.argsort(-indeg, kind="stable")[:n],
                  "oracle-best": np.argsort(-freq, kind="stable")[:n]}
        for name, members in orders.items():
            inset = np.zeros(N, bool)
            inset[members] = True
            hit = inset[reads]
            by_hop = {f"hops {lo}-{hi - 1}": round(float(hit[(hops >= lo) & (hops < hi)].mean()), 4) for lo, hi in ((1, 4), (4, 16), (16, 64), (64, 1000))
                      if ((hops >= lo) & (hops < hi)).any()}
            out["orders"][f"{name} @ {int(share * 100)} %"] = {"hit_rate": round(float(hit.mean()), 4), "by_hop": by_hop}
            log(f"[study] {name:12s} n = {share:4.0%} of N: {hit.mean():6.2%} of the row reads served from HBM   {by_hop}")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
