"""Index construction at 10-100 M points on one MI355X (the reference delegates it to DiskANN's
``build_disk_index``: root README.md:46-58, BANG_Base/ReadMe.pdf p.1-2).

`synth.build_graph` finds exact nearest neighbours by brute force, which stops being practical beyond a few million points
(the selection over N^2 distances, not the matmul).  This module builds a Vamana-STYLE graph instead:

1. candidate neighbours by a partitioned search on the GPU: coarse k-means (C ~ sqrt(N) cells), every point is compared with
   the points of its own and the `probes` nearest cells (one [cell x candidates] matmul + top-k per cell);
2. DiskANN's robust prune (the alpha rule of Vamana) over each point's candidate list, fully batched: a candidate c is dropped
   once a kept neighbour p* satisfies alpha * d(p*, c) <= d(p, c); at most R/2 survive;
3. the remaining slots are filled with uniformly random long-range links (what keeps greedy search short on a graph that was
   not grown incrementally), adjacency sorted ascending as bang_preprocess.py:102-104 leaves it.

The PQ side (DiskANN's layout: global centroid, D split into m chunks, 256 pivots per chunk) is `synth.train_pq`.

Everything is tooling around the search path: it produces the index FILES' contents; nothing here runs during a query."""
from __future__ import annotations

import time

import numpy as np
import torch

from . import synth
from .formats import Index, pack_graph


def _kmeans(x: torch.Tensor, C: int, iters: int, g: torch.Generator, sample: int = 262144):
    N = x.shape[0]
    s = x[torch.randperm(N, generator=g, device=x.device)[: min(N, sample)]]
    cen = s[torch.randperm(s.shape[0], generator=g, device=x.device)[:C]].clone()
    for _ in range(iters):
        lab = _assign(s, cen)
        sums = torch.zeros_like(cen).index_add_(0, lab, s)
        cnt = torch.zeros(C, device=x.device).index_add_(0, lab, torch.ones(s.shape[0], device=x.device))
        nz = cnt > 0
        cen[nz] = sums[nz] / cnt[nz, None]
    return cen


def _assign(x: torch.Tensor, cen: torch.Tensor, block: int = 1 << 18) -> torch.Tensor:
    cn = (cen * cen).sum(1)
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    for a in range(0, x.shape[0], block):
        b = x[a:a + block]
        out[a:a + block] = (cn[None, :] - 2.0 * (b @ cen.T)).argmin(dim=1)
    return out


def candidate_neighbours(x: torch.Tensor, K: int, probes: int = 6, seed: int = synth.SEED, cells: int = 0, log=None):
    """Approximate K nearest neighbours of every point (excluding itself) through a coarse partition.
    Returns (ids int64 [N, K], squared distances f32 [N, K]) sorted ascending."""
    N, D = x.shape
    dev = x.device
    g = synth._gen(seed + 11, dev)
    C = cells or int(max(16, min(8192, round((N ** 0.5) / 1.5))))
    cen = _kmeans(x, C, 6, g)
    lab = _assign(x, cen)
    order = torch.argsort(lab, stable=True)
    counts = torch.bincount(lab, minlength=C)
    starts = torch.cumsum(counts, 0) - counts
    cd = torch.cdist(cen, cen)
    near = torch.topk(cd, min(probes, C), dim=1, largest=False).indices           # [C, probes], self first
    xn = (x * x).sum(1)
    out_i = torch.empty((N, K), dtype=torch.int64, device=dev)
    out_d = torch.empty((N, K), dtype=torch.float32, device=dev)
    starts_h, counts_h, near_h = starts.tolist(), counts.tolist(), near.tolist()
    t0 = time.time()
    for c in range(C):
        if counts_h[c] == 0:
            continue
        rows = order[starts_h[c]: starts_h[c] + counts_h[c]]
        cand = torch.cat([order[starts_h[j]: starts_h[j] + counts_h[j]] for j in near_h[c] if counts_h[j] > 0])
        if cand.shape[0] <= K:                                   # tiny neighbourhood: widen to a random sample
            extra = torch.randint(0, N, (4 * K,), generator=g, device=dev)
            cand = torch.unique(torch.cat([cand, extra]))
        xr = x[rows]
        for a in range(0, rows.shape[0], 8192):                  # bound the distance tile
            r = rows[a:a + 8192]
            d = xn[r][:, None] + xn[cand][None, :] - 2.0 * (xr[a:a + 8192] @ x[cand].T)
            d.masked_fill_(r[:, None] == cand[None, :], float("inf"))
            dd, ii = torch.topk(d, K, dim=1, largest=False)
            out_i[r] = cand[ii]
            out_d[r] = dd.clamp_min_(0.0)
        if log and (c + 1) % 512 == 0:
            log(f"[build] candidate search: cell {c + 1}/{C} ({time.time() - t0:.1f}s)")
    return out_i, out_d


def robust_prune(x: torch.Tensor, cand_i: torch.Tensor, cand_d: torch.Tensor, keep: int, alpha: float = 1.2, block: int = 32768):
    """DiskANN / Vamana robust prune, batched.  cand_* [N, K] sorted ascending by distance to the point.  Walks the list in order;
    a candidate that is still alive is KEPT and kills every later candidate c with alpha * d(kept, c) <= d(point, c) (distances, not
    squares: the comparison is done as alpha^2 * d2 <= d2').  Returns (mask bool [N, K] of kept candidates, at most `keep` per row)."""
    N, K = cand_i.shape
    out = torch.zeros((N, K), dtype=torch.bool, device=x.device)
    a2 = alpha * alpha
    for s in range(0, N, block):
        ci = cand_i[s:s + block]
        dp = cand_d[s:s + block]                                  # d2(point, c)
        v = x[ci]                                                 # [B, K, D]
        n2 = (v * v).sum(2)
        dcc = (n2[:, :, None] + n2[:, None, :] - 2.0 * torch.bmm(v, v.transpose(1, 2))).clamp_min_(0.0)   # [B, K, K] d2(c_i, c_j)
        alive = torch.isfinite(dp)
        kept = torch.zeros_like(alive)
        n_kept = torch.zeros(ci.shape[0], dtype=torch.int64, device=x.device)
        for k in range(K):
            take = alive[:, k] & (n_kept < keep)
            kept[:, k] = take
            n_kept += take.to(torch.int64)
            kill = take[:, None] & (a2 * dcc[:, k, :] <= dp)
            alive &= ~kill
        out[s:s + block] = kept
    return out


def build_graph_large(x: torch.Tensor, R: int, seed: int = synth.SEED, K: int = 48, probes: int = 6, alpha: float = 1.2, log=None):
    """(degrees int64 [N], adjacency int64 [N, R]): <= R/2 robust-pruned near neighbours + random long-range links up to R,
    distinct, != self, sorted ascending; the tail of a short row is 0."""
    N = x.shape[0]
    dev = x.device
    t0 = time.time()
    ci, cd = candidate_neighbours(x, K, probes=probes, seed=seed, log=log)
    if log:
        log(f"[build] {K} candidate neighbours per point in {time.time() - t0:.1f}s")
    t0 = time.time()
    kept = robust_prune(x, ci, cd, keep=R // 2, alpha=alpha)
    if log:
        log(f"[build] robust prune (alpha {alpha}): {float(kept.sum(1).float().mean()):.1f} of {K} kept on average, {time.time() - t0:.1f}s")
    g = synth._gen(seed + 1, dev)
    near = torch.where(kept, ci, torch.full_like(ci, -1))
    # per row: the kept neighbours first, then random links; R entries in all
    near_sorted, _ = torch.sort(near, dim=1, descending=True)                    # valid ids first, -1 at the end
    near_sorted = near_sorted[:, : R // 2]
    rnd = torch.randint(0, N, (N, R), generator=g, device=dev)
    n_near = (near_sorted >= 0).sum(1, keepdim=True)
    col = torch.arange(R, device=dev)[None, :]
    pad = torch.full((N, R - near_sorted.shape[1]), -1, dtype=torch.int64, device=dev)
    adj = torch.where(col < n_near, torch.cat([near_sorted, pad], 1), rnd)
    adj, _ = torch.sort(adj, dim=1)
    self_id = torch.arange(N, device=dev)[:, None]
    ok = torch.ones_like(adj, dtype=torch.bool)
    ok[:, 1:] = adj[:, 1:] != adj[:, :-1]
    ok &= adj != self_id
    deg = ok.sum(dim=1)
    pos = torch.cumsum(ok.to(torch.int64), dim=1) - 1
    out = torch.zeros_like(adj)
    rows = torch.arange(N, device=dev)[:, None].expand_as(adj)
    out[rows[ok], pos[ok]] = adj[ok]
    return deg, out


def make_index_large(N: int, D: int, dtype: str, R: int, m: int, Q: int, K: int = 10, n_clusters: int = 1024,
                     seed: int = synth.SEED, device="cuda", pq_iters: int = 6, log=None):
    """A structured index of N >= a few million points with brute-force ground truth for Q queries.
    Returns (Index, queries np [Q, D], gt_ids np u32 [Q, K], gt_dists np f32 [Q, K])."""
    t0 = time.time()
    x = synth.make_vectors(N, D, dtype, n_clusters=n_clusters, seed=seed, device=device)
    deg, adj = build_graph_large(x, R, seed=seed, log=log)
    medoid = int(synth._sq_norms(x - x.mean(dim=0)).argmin())
    deg_np = deg.cpu().numpy().astype(np.uint32)
    adj_np = adj.cpu().numpy().astype(np.uint32)
    del deg, adj
    torch.cuda.empty_cache() if str(device).startswith("cuda") else None
    t1 = time.time()
    pivots, centroid, off, codes = synth.train_pq(x, m, iters=pq_iters, seed=seed)
    if log:
        log(f"[build] PQ: {m} chunks trained and {N} points encoded in {time.time() - t1:.1f}s")
    q = synth.make_queries(x, Q, dtype, seed=seed)
    t1 = time.time()
    gt_i, gt_d = synth.knn(x, q, K, row_block=2048)
    if log:
        log(f"[build] brute-force ground truth for {Q} queries in {time.time() - t1:.1f}s")
    vec_np = synth.to_numpy(x, dtype)
    del x
    graph = pack_graph(vec_np, deg_np, adj_np)
    ix = Index(dtype=dtype, N=N, D=D, R=R, m=m, medoid=medoid, graph=graph, codes=codes.cpu().numpy(),
               pivots=pivots.cpu().numpy().astype(np.float32), centroid=centroid.cpu().numpy().astype(np.float32), chunk_off=off)
    if log:
        log(f"[build] index of {N} points built in {time.time() - t0:.1f}s")
    return ix, synth.to_numpy(q, dtype), gt_i.cpu().numpy().astype(np.uint32), gt_d.cpu().numpy().astype(np.float32)
