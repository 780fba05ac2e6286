"""Index construction at 10-100 M points on one MI355X (the reference delegates it to DiskANN's
``build_disk_index -R 64 -L 200``: root README.md:46-58, BANG_Base/ReadMe.pdf p.1-2).

`synth.build_graph` finds exact nearest neighbours by brute force, which stops being practical beyond a few million points
(the selection over N^2 distances, not the matmul).  This module builds a Vamana-STYLE graph in batched passes instead:

1. candidate neighbours by a partitioned search on the GPU: a two-level k-means partition (cells of ~2 K points), every point
   is compared with the points of its own and the `probes` nearest cells (one [cell x candidates] matmul + top-k per cell;
   uint8 / int8 data is held as bfloat16, in which its values, their products and the f32-accumulated dot products are exact);
2. DiskANN's robust prune (the alpha rule of Vamana) over each point's candidate list, fully batched: a candidate c is dropped
   once a kept neighbour p* satisfies alpha * d(p*, c) <= d(p, c);
3. REVERSE edges (Vamana inserts p into the list of every neighbour it selects): every kept edge p -> c offers p to c; a point's
   kept and offered neighbours together go through a SECOND robust prune.  This is what makes the graph navigable from any
   direction -- a one-pass pruned kNN graph leaves many points with few or no inbound near edges;
4. the remaining slots (at least `n_random`) are filled with uniformly random long-range links (the shortcut edges Vamana gets
   from inserting points along greedy-search paths from the medoid), adjacency sorted ascending as bang_preprocess.py:102-104
   leaves it.

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
    s = x[torch.randperm(N, generator=g, device=x.device)[: min(N, sample)]].float()
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
    cw = cen.to(x.dtype)
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    for a in range(0, x.shape[0], block):
        b = x[a:a + block]
        out[a:a + block] = (cn[None, :] - 2.0 * (b @ cw.T).float()).argmin(dim=1)
    return out


class Partition:
    """A two-level k-means partition: `order` lists the points cell by cell (cells of one top-level cell are consecutive), cell c
    owns order[starts[c] : starts[c] + counts[c]], `near` [C, n] are each cell's nearest cells (itself first), cell c belongs to
    top-level cell top_of[c], top-level cell t owns order[tstarts[t] : tstarts[t] + tcounts[t]], `tnear` its nearest top cells."""
    pass


def partition(x: torch.Tensor, cell: int, g: torch.Generator, n_near: int = 64, cells: int = 0) -> Partition:
    """One level of C ~ N / cell centroids would cost an [N x C] distance matrix (N = 1e8: 5e12 entries); two levels cost
    N x (C1 + C2) with C1 ~ C2 ~ sqrt(C)."""
    N = x.shape[0]
    dev = x.device
    C = cells or max(1, int(round(N / cell)))
    P = Partition()
    if C <= 1024:
        cen = _kmeans(x, max(1, min(C, N)), 6, g)
        lab = _assign(x, cen)
        top_of = torch.zeros(cen.shape[0], dtype=torch.int64, device=dev)
        cen1 = cen.mean(0, keepdim=True)
    else:
        C1 = int(max(16, min(1024, round(C ** 0.5))))
        cen1 = _kmeans(x, C1, 6, g)
        lab1 = _assign(x, cen1)
        order1 = torch.argsort(lab1, stable=True)
        counts1 = torch.bincount(lab1, minlength=C1).tolist()
        lab = torch.empty(N, dtype=torch.int64, device=dev)
        cens, tops, base, a = [], [], 0, 0
        for c in range(C1):
            n = counts1[c]
            if n == 0:
                continue
            rows = order1[a:a + n]
            a += n
            k = max(1, min(int(round(n / cell)), n))
            xs = x[rows]
            cen_c = _kmeans(xs, k, 4, g, sample=65536)
            lab[rows] = _assign(xs, cen_c) + base
            cens.append(cen_c)
            tops.append(torch.full((k,), c, dtype=torch.int64, device=dev))
            base += k
        cen = torch.cat(cens)
        top_of = torch.cat(tops)
        del lab1, order1
    Cn = cen.shape[0]
    P.C, P.cen, P.lab, P.top_of = Cn, cen, lab, top_of
    P.order = torch.argsort(lab, stable=True)
    P.counts = torch.bincount(lab, minlength=Cn)
    P.starts = torch.cumsum(P.counts, 0) - P.counts
    nn = min(n_near, Cn)
    P.near = torch.empty((Cn, nn), dtype=torch.int64, device=dev)
    for a in range(0, Cn, 4096):                                                   # self first
        P.near[a:a + 4096] = torch.topk(torch.cdist(cen[a:a + 4096], cen), nn, dim=1, largest=False).indices
    T = cen1.shape[0]
    P.T = T
    P.tcounts = torch.zeros(T, dtype=torch.int64, device=dev).index_add_(0, top_of, P.counts)
    P.tstarts = torch.cumsum(P.tcounts, 0) - P.tcounts
    P.tnear = torch.topk(torch.cdist(cen1, cen1), min(16, T), dim=1, largest=False).indices
    return P


def smallworld_links(P: Partition, first: int, count: int, n_links: int, N: int, g: torch.Generator) -> torch.Tensor:
    """Long-range links for the points [first, first + count): n_links targets per point drawn SCALE-uniformly -- link j picks one of
    seven nested neighbourhoods of its point (its 4 / 16 / 64 nearest cells, its top-level cell, the 4 / 16 nearest top-level cells,
    everything) and a uniformly random member of it.  Equal link mass per distance scale is what makes greedy routing short
    (Kleinberg's small-world construction; Vamana gets such edges from greedy-search paths while it inserts points): uniform random
    links alone leave a 1e8-point graph with nothing between 'anywhere' and 'the next 48 neighbours'."""
    dev = P.order.device
    cells = P.lab[first:first + count]                                            # [B]
    tops = P.top_of[cells]
    B = count
    out = torch.empty((B, n_links), dtype=torch.int64, device=dev)
    u = torch.rand((B, n_links, 2), generator=g, device=dev)
    nn, nt = P.near.shape[1], P.tnear.shape[1]
    for j in range(n_links):
        sc = j % 7
        if sc < 3:                                                                 # among the 4 / 16 / 64 nearest cells
            width = min(nn, 4 ** (sc + 1))
            tc = P.near[cells, (u[:, j, 0] * width).long().clamp_(max=width - 1)]
            out[:, j] = P.order[P.starts[tc] + (u[:, j, 1] * P.counts[tc]).long().minimum(P.counts[tc] - 1).clamp_min_(0)]
        elif sc < 6:                                                               # own top-level cell, 4 / 16 nearest top-level cells
            width = min(nt, 4 ** (sc - 3))
            tt = P.tnear[tops, (u[:, j, 0] * width).long().clamp_(max=width - 1)]
            out[:, j] = P.order[P.tstarts[tt] + (u[:, j, 1] * P.tcounts[tt]).long().minimum(P.tcounts[tt] - 1).clamp_min_(0)]
        else:
            out[:, j] = (u[:, j, 1] * N).long().clamp_(max=N - 1)
    return out


def candidate_neighbours(x: torch.Tensor, K: int, probes: int = 6, seed: int = synth.SEED, cells: int = 0, log=None, cell: int = 2048,
                         idx_dtype=torch.int64, want_partition: bool = False, select: str = "topk", state=None, cell_range=None):
    """Approximate K nearest neighbours of every point (excluding itself) through a coarse partition.
    Returns (ids [N, K] of idx_dtype, squared distances f32 [N, K]) sorted ascending (+ the Partition if asked for).
    x may be f32 or (exact for 8-bit data) bf16.
    SLICED use (an index whose [N, K] tables do not fit HBM beside the vectors): `state` = a dict that carries the partition, the
    norms and the random stream from slice to slice, `cell_range` = (c0, c1) -- the tables then hold the points of those cells only, in
    the partition's order: row j belongs to point P.order[P.starts[c0] + j].  Slice by slice the results equal the one-shot call's."""
    N, D = x.shape
    dev = x.device
    if state is not None and "P" in state:
        P, g, xn = state["P"], state["g"], state["xn"]
    else:
        g = synth._gen(seed + 11, dev)
        t0 = time.time()
        P = partition(x, cell, g, cells=cells)
        if log:
            log(f"[build] partition: {P.C} cells of ~{N // max(1, P.C)} points in {time.time() - t0:.1f}s")
        xn = (x.float() * x.float()).sum(1) if x.dtype == torch.float32 else torch.cat([(x[a:a + (1 << 22)].float() ** 2).sum(1) for a in range(0, N, 1 << 22)])
        if state is not None:
            state.update(P=P, g=g, xn=xn)
    C = P.C
    order = P.order
    starts_h, counts_h, near_h = P.starts.tolist(), P.counts.tolist(), P.near[:, :probes].tolist()
    c0, c1 = cell_range if cell_range is not None else (0, C)
    base = starts_h[c0] if c0 < C else N
    n_out = N if cell_range is None else (starts_h[c1 - 1] + counts_h[c1 - 1] - base if c1 > c0 else 0)
    out_i = torch.empty((n_out, K), dtype=idx_dtype, device=dev)
    out_d = torch.empty((n_out, K), dtype=torch.float32, device=dev)
    t0 = time.time()
    step = max(512, C // 8)
    for c in range(c0, c1):
        if counts_h[c] == 0:
            continue
        rows = order[starts_h[c]: starts_h[c] + counts_h[c]]
        cand = torch.cat([order[starts_h[j]: starts_h[j] + counts_h[j]] for j in near_h[c] if counts_h[j] > 0])
        if cand.shape[0] <= K:                                   # tiny neighbourhood: widen to a random sample
            extra = torch.randint(0, N, (4 * K,), generator=g, device=dev)
            cand = torch.unique(torch.cat([cand, extra]))
        xc = x[cand].float().T.contiguous()                      # (bfloat16 is the STORAGE type of 8-bit data: a bf16 x bf16 product
        xnc = xn[cand]                                           # would come back rounded to bf16 -- the arithmetic is f32, exact)
        for a in range(0, rows.shape[0], 8192):                  # bound the distance tile
            r = rows[a:a + 8192]
            # |c|^2 - 2 p.c: the point's own |p|^2 is constant along a row -- it joins the K selected values only (one pass less over the tile)
            d = torch.addmm(xnc[None, :], x[r].float(), xc, alpha=-2.0)
            if select == "topk":
                d.masked_fill_(r[:, None] == cand[None, :], float("inf"))
                dd, ii = torch.topk(d, K, dim=1, largest=False)
            else:
                # "groupmin": the candidates are dealt round-robin into G = 2K groups and every group keeps its nearest member: ONE
                # pass over the distance tile instead of a top-k selection (3x faster at 12 K candidates).  The K nearest of those
                # 2K winners are the point's candidates: a true near neighbour is lost only when a nearer one sits in its group
                # (~5 % of the ten nearest), and the robust prune drops half the list anyway.
                G = 2 * K
                nc = d.shape[1]
                npad = (nc + G - 1) // G * G
                if npad != nc:
                    d = torch.cat([d, torch.full((d.shape[0], npad - nc), float("inf"), device=dev)], 1)
                gv, ga = d.view(d.shape[0], npad // G, G).min(dim=1)                     # [rows, G]
                gi = (ga * G + torch.arange(G, device=dev)[None, :]).clamp_(max=nc - 1)
                gv = torch.where(cand[gi] == r[:, None], torch.full_like(gv, float("inf")), gv)   # (the point itself won its group)
                dd, o = torch.sort(gv, dim=1)
                dd, ii = dd[:, :K], torch.gather(gi, 1, o[:, :K])
            if cell_range is not None:                               # (the points of a cell are consecutive in the partition's order)
                j0 = starts_h[c] - base + a
                out_i[j0:j0 + r.shape[0]] = cand[ii].to(idx_dtype)
                out_d[j0:j0 + r.shape[0]] = (dd + xn[r][:, None]).clamp_min_(0.0)
            else:
                out_i[r] = cand[ii].to(idx_dtype)
                out_d[r] = (dd + xn[r][:, None]).clamp_min_(0.0)
        if log and (c + 1) % step == 0:
            log(f"[build] candidate search: cell {c + 1}/{C} ({time.time() - t0:.1f}s)")
    return (out_i, out_d, P) if want_partition else (out_i, out_d)


def robust_prune(x: torch.Tensor, cand_i: torch.Tensor, cand_d: torch.Tensor, keep: int, alpha: float = 1.2, block: int = 0):
    """DiskANN / Vamana robust prune, batched.  cand_* [N, K] sorted ascending by distance to the point (padding: distance +inf).
    Walks the list in order; a candidate that is still alive is KEPT and kills every later candidate c with
    alpha * d(kept, c) <= d(point, c) (distances, not squares: the comparison is done as alpha^2 * d2 <= d2').
    Returns (mask bool [N, K] of kept candidates, at most `keep` per row)."""
    N, K = cand_i.shape
    out = torch.zeros((N, K), dtype=torch.bool, device=x.device)
    a2 = alpha * alpha
    if block <= 0:                                                # the K-step loop below is launch bound: big blocks where memory allows
        block = 262144 if (x.is_cuda and N >= (1 << 22)) else 32768
    for s in range(0, N, block):
        ci = cand_i[s:s + block].long()
        dp = cand_d[s:s + block]                                  # d2(point, c)
        v = x[ci.clamp_min(0)].float()                            # [B, K, D]
        n2 = (v * v).sum(2)
        dcc = torch.baddbmm(n2[:, None, :], v, v.transpose(1, 2), alpha=-2.0).add_(n2[:, :, None]).clamp_min_(0.0)   # [B, K, K] d2(c_i, c_j)
        del v
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


def reverse_slots(near_i: torch.Tensor, slots: int = 24, block: int = 1 << 20) -> torch.Tensor:
    """rev [N, slots]: the offers every point received (see reverse_candidates), -1 = empty"""
    N = near_i.shape[0]
    dev = near_i.device
    rev = torch.full((N * slots,), -1, dtype=(torch.int32 if N < (1 << 31) else torch.int64), device=dev)   # (2e8 points x 24 slots: 19 GB instead of 38)
    for s in range(0, N, block):
        c = near_i[s:s + block].long()
        p = torch.arange(s, s + c.shape[0], device=dev)[:, None].expand_as(c)
        ok = c >= 0
        cc, pp = c[ok], p[ok]
        slot = ((pp * 0x9E3779B1 + cc * 0x85EBCA77) >> 7) % slots
        rev.scatter_reduce_(0, cc * slots + slot, pp.to(rev.dtype), reduce="amax", include_self=True)
    return rev.view(N, slots)


def reverse_candidates(x: torch.Tensor, near_i: torch.Tensor, near_d: torch.Tensor, slots: int = 24, block: int = 1 << 20, rev=None, row_range=None):
    """Every kept edge p -> c offers p to c (Vamana's reverse insertion).  near_i [N, K1] (-1 = empty), near_d [N, K1].
    A point keeps at most `slots` offers: an offer lands in slot hash(p, c) % slots and the largest p wins a contested slot
    (scatter-max: deterministic, no global sort of the ~N x 30 edges).  Returns the merged candidate lists -- own kept + offered,
    distinct, sorted ascending by distance, padding id -1 / distance +inf: (ids [N, K1 + slots] of near_i.dtype, d2 f32)."""
    N, K1 = near_i.shape
    dev = x.device
    if rev is None:
        rev = reverse_slots(near_i, slots, block)
    r0, r1 = row_range if row_range is not None else (0, N)          # (sliced use: the merged lists of the points [r0, r1) only)
    out_i = torch.empty((r1 - r0, K1 + slots), dtype=near_i.dtype, device=dev)
    out_d = torch.empty((r1 - r0, K1 + slots), dtype=torch.float32, device=dev)
    for s in range(r0, r1, 1 << 18):
        r = rev[s:min(s + (1 << 18), r1)].long()
        B = r.shape[0]
        xr = x[r.clamp_min(0)].float()                            # [B, slots, D]
        xp = x[s:s + B].float()
        rd = ((xr - xp[:, None, :]) ** 2).sum(2)
        rd = torch.where(r >= 0, rd, torch.full_like(rd, float("inf")))
        ids = torch.cat([near_i[s:s + B].long(), r], 1)
        dd = torch.cat([torch.where(near_i[s:s + B] >= 0, near_d[s:s + B], torch.full_like(near_d[s:s + B], float("inf"))), rd], 1)
        # distinct: sort by id, drop repeats (an offered point may already be a kept neighbour), then sort by distance
        ids_s, o = torch.sort(ids, dim=1)
        dd_s = torch.gather(dd, 1, o)
        dup = torch.zeros_like(ids_s, dtype=torch.bool)
        dup[:, 1:] = ids_s[:, 1:] == ids_s[:, :-1]
        dd_s = torch.where(dup | (ids_s < 0), torch.full_like(dd_s, float("inf")), dd_s)
        dd_o, o2 = torch.sort(dd_s, dim=1, stable=True)
        ids_o = torch.gather(ids_s, 1, o2)
        ids_o = torch.where(torch.isfinite(dd_o), ids_o, torch.full_like(ids_o, -1))
        out_i[s - r0:s - r0 + B] = ids_o.to(near_i.dtype)
        out_d[s - r0:s - r0 + B] = dd_o
    return out_i, out_d


def _compact_kept(ci: torch.Tensor, cd: torch.Tensor, kept: torch.Tensor, width: int):
    """The kept candidates of every row moved to the front (order preserved): (ids [N, width] with -1 padding, d2 with +inf)."""
    N, K = ci.shape
    out_i = torch.full((N, width), -1, dtype=ci.dtype, device=ci.device)
    out_d = torch.full((N, width), float("inf"), dtype=torch.float32, device=ci.device)
    for s in range(0, N, 1 << 20):
        k = kept[s:s + (1 << 20)]
        pos = torch.cumsum(k.to(torch.int64), 1) - 1
        ok = k & (pos < width)
        rows = torch.arange(k.shape[0], device=ci.device)[:, None].expand_as(k)
        oi, od = out_i[s:s + (1 << 20)], out_d[s:s + (1 << 20)]
        oi[rows[ok], pos[ok]] = ci[s:s + (1 << 20)][ok]
        od[rows[ok], pos[ok]] = cd[s:s + (1 << 20)][ok]
    return out_i, out_d


def _count_true(mask: torch.Tensor, block: int = 1 << 22) -> int:
    """number of True entries, block by block (a reduction over a bool tensor first widens ALL of it to int64: 83 GiB for 2e8 x 56)"""
    return sum(int(torch.count_nonzero(mask[a:a + block])) for a in range(0, mask.shape[0], block))


def _finish_rows(near_i: torch.Tensor, P: Partition, R: int, N: int, seed: int, links: str, idx_dtype):
    """the near neighbours of every point, then random long-range links up to R entries; distinct, != self, sorted ascending"""
    dev = near_i.device
    g = synth._gen(seed + 1, dev)
    adj_out = torch.zeros((N, R), dtype=idx_dtype, device=dev)
    deg_out = torch.empty(N, dtype=torch.int64, device=dev)
    W = near_i.shape[1]
    for s in range(0, N, 1 << 20):                               # per row: the near neighbours, then random links; R entries in all
        near = near_i[s:s + (1 << 20)].long()
        B = near.shape[0]
        rnd = smallworld_links(P, s, B, R, N, g) if links == "smallworld" else torch.randint(0, N, (B, R), generator=g, device=dev)
        n_n = (near >= 0).sum(1, keepdim=True)
        col = torch.arange(R, device=dev)[None, :]
        pad = torch.full((B, R - W), -1, dtype=torch.int64, device=dev)
        adj = torch.where(col < n_n, torch.cat([near, pad], 1), rnd)
        adj, _ = torch.sort(adj, dim=1)
        self_id = torch.arange(s, s + B, device=dev)[:, None]
        ok = torch.ones_like(adj, dtype=torch.bool)
        ok[:, 1:] = adj[:, 1:] != adj[:, :-1]
        ok &= adj != self_id
        deg_out[s:s + B] = ok.sum(dim=1)
        pos = torch.cumsum(ok.to(torch.int64), dim=1) - 1
        out = torch.zeros_like(adj)
        rows = torch.arange(B, device=dev)[:, None].expand_as(adj)
        out[rows[ok], pos[ok]] = adj[ok]
        adj_out[s:s + B] = out.to(idx_dtype)
    return deg_out, adj_out


def _build_graph_sliced(holder: list, R, seed, K, probes, alpha, log, reverse, n_random, cell, cells, idx_dtype, links, select, slices, narrow):
    """build_graph_large with the big per-point tables held for one slice of the points at a time (see there)."""
    x = holder[0]
    N = x.shape[0]
    dev = x.device
    n_near = R - n_random
    keep1 = min(n_near, R // 2)
    st = {}
    t0 = time.time()
    near_i = near_d = None
    kept_total = 0
    for sl in range(slices):
        if sl == 0:                                              # (the partition is made by the first call)
            ci, cd, P = candidate_neighbours(x, K, probes=probes, seed=seed, log=log, cell=cell, cells=cells, idx_dtype=idx_dtype, want_partition=True,
                                             select=select, state=st, cell_range=(0, 0))
            del ci, cd
            C = P.C
            near_i = torch.full((N, keep1), -1, dtype=idx_dtype, device=dev)
            near_d = torch.full((N, keep1), float("inf"), dtype=torch.float32, device=dev)
        c0, c1 = C * sl // slices, C * (sl + 1) // slices
        if c1 <= c0:
            continue
        ci, cd = candidate_neighbours(x, K, probes=probes, seed=seed, log=None, cell=cell, cells=cells, idx_dtype=idx_dtype, select=select, state=st,
                                      cell_range=(c0, c1))
        kept = robust_prune(x, ci, cd, keep=keep1, alpha=alpha)
        kept_total += _count_true(kept)
        ni, nd = _compact_kept(ci, cd, kept, keep1)
        del ci, cd, kept
        base = int(P.starts[c0])
        rows = P.order[base:base + ni.shape[0]]
        near_i[rows] = ni
        near_d[rows] = nd
        del ni, nd, rows
        if x.is_cuda:
            torch.cuda.empty_cache()
        if log:
            log(f"[build] slice {sl + 1}/{slices}: candidates + robust prune of cells [{c0}, {c1}) ({time.time() - t0:.1f}s)")
    if log:
        log(f"[build] {K} candidate neighbours per point, robust prune (alpha {alpha}): {kept_total / N:.1f} kept on average, {time.time() - t0:.1f}s")
    st.pop("xn", None)
    if narrow in ("uint8", "int8") and x.dtype != torch.float32:
        # the matmuls are done: from here on only `x[...].float()` gathers follow -- the bf16 copy (2 bytes per value) makes room
        x8 = torch.empty(x.shape, dtype=(torch.uint8 if narrow == "uint8" else torch.int8), device=dev)
        for a in range(0, N, 1 << 24):
            x8[a:a + (1 << 24)] = x[a:a + (1 << 24)].to(x8.dtype)
        holder[0] = x8
        del x
        x = x8
        if x.is_cuda:
            torch.cuda.empty_cache()
    if reverse:
        t0 = time.time()
        slots = max(8, R // 2 - 8)
        rev = reverse_slots(near_i, slots)
        near2 = torch.full((N, n_near), -1, dtype=idx_dtype, device=dev)
        kept2_total = 0
        for sl in range(slices * 3):                             # (the merged lists are wider than the candidate tables, and the final lists are alive beside them)
            r0, r1 = N * sl // (slices * 3), N * (sl + 1) // (slices * 3)
            mi, md = reverse_candidates(x, near_i, near_d, slots=slots, rev=rev, row_range=(r0, r1))
            kept2 = robust_prune(x, mi, md, keep=n_near, alpha=alpha)
            kept2_total += _count_true(kept2)
            n2, _ = _compact_kept(mi, md, kept2, n_near)
            near2[r0:r1] = n2
            del mi, md, kept2, n2
            if x.is_cuda:
                torch.cuda.empty_cache()
        del rev, near_i, near_d
        near_i = near2
        if x.is_cuda:
            torch.cuda.empty_cache()
        if log:
            log(f"[build] reverse edges + second prune: {kept2_total / N:.1f} near neighbours per point, {time.time() - t0:.1f}s")
    else:
        del near_d
    return _finish_rows(near_i, P, R, N, seed, links, idx_dtype)


def build_graph_large(x: torch.Tensor, R: int, seed: int = synth.SEED, K: int = 48, probes: int = 6, alpha: float = 1.2, log=None,
                      reverse: bool = True, n_random: int = 12, cell: int = 2048, cells: int = 0, idx_dtype=torch.int64,
                      links: str = "smallworld", select: str = "topk", slices: int = 0, narrow: str | None = None):
    """(degrees int64 [N], adjacency [N, R] of idx_dtype): robust-pruned near neighbours (two passes with reverse edges in
    between; at most R - n_random of them) + random long-range links up to R, distinct, != self, sorted ascending; the tail of a
    short row is 0.  reverse=False, n_random=R//2: the one-pass builder of round 2.
    x: the vectors, or a one-element LIST holding them (sliced build: the builder swaps the bf16 tensor for its 8-bit original once
    the matmuls are done -- narrow = "uint8" / "int8" -- and the caller finds the swapped tensor in the list).
    slices > 1 (0 = auto: 1 up to 2e8 points, else ceil(N / 1e8)): the [N, 48] candidate tables and the [N, 56] merged lists exist
    for one slice of the points at a time (2e8 points peak at 224 GiB of HBM in one piece; 3e8 in slices at ~210).  Same graph as the
    one-shot build (tests/test_build.py)."""
    holder = x if isinstance(x, list) else None
    if holder is not None:
        x = holder[0]
    N = x.shape[0]
    dev = x.device
    if slices <= 0:
        slices = 1 if N <= 200_000_000 else -(-N // 100_000_000)
    if slices > 1:
        if holder is None:
            holder = [x]
        del x
        return _build_graph_sliced(holder, R, seed, K, probes, alpha, log, reverse, n_random, cell, cells, idx_dtype, links, select,
                                   slices, narrow)
    t0 = time.time()
    ci, cd, P = candidate_neighbours(x, K, probes=probes, seed=seed, log=log, cell=cell, cells=cells, idx_dtype=idx_dtype, want_partition=True,
                                     select=select)
    if log:
        log(f"[build] {K} candidate neighbours per point in {time.time() - t0:.1f}s")
    t0 = time.time()
    n_near = R - n_random
    keep1 = min(n_near, R // 2)
    kept = robust_prune(x, ci, cd, keep=keep1, alpha=alpha)
    if log:
        log(f"[build] robust prune (alpha {alpha}): {_count_true(kept) / N:.1f} of {K} kept on average, {time.time() - t0:.1f}s")
    near_i, near_d = _compact_kept(ci, cd, kept, keep1)
    del ci, cd, kept
    if x.is_cuda:
        torch.cuda.empty_cache()                                 # (the next tables are of other shapes: cached blocks would only fragment the pool)
    if reverse:
        t0 = time.time()
        mi, md = reverse_candidates(x, near_i, near_d, slots=max(8, R // 2 - 8))
        del near_i, near_d
        if x.is_cuda:
            torch.cuda.empty_cache()
        kept2 = robust_prune(x, mi, md, keep=n_near, alpha=alpha)
        near_i, near_d = _compact_kept(mi, md, kept2, n_near)
        if log:
            log(f"[build] reverse edges + second prune: {_count_true(kept2) / N:.1f} near neighbours per point, {time.time() - t0:.1f}s")
        del mi, md, kept2, near_d
        if x.is_cuda:
            torch.cuda.empty_cache()
    return _finish_rows(near_i, P, R, N, seed, links, idx_dtype)


def pack_graph_device(x: torch.Tensor, dtype: str, deg: torch.Tensor, adj: torch.Tensor, block: int = 1 << 20) -> np.ndarray:
    """formats.pack_graph for a large index: the [T vec[D]][u32 degree][u32 id x R] entries are assembled on the device block by
    block and land in ONE host array (N x entryLen bytes: 38.8 GB for 1e8 SIFT-like points) -- no [N, R] int64 host copies.  The
    blocks come down into pinned buffers (asynchronously, two in flight) and a few threads move them into the array."""
    from concurrent.futures import ThreadPoolExecutor
    N, D = x.shape
    R = adj.shape[1]
    isz = 4 if dtype == "float" else 1
    el = D * isz + 4 + 4 * R
    out = np.empty((N, el), dtype=np.uint8)
    cuda = x.is_cuda
    nbuf = 4
    pins = [torch.empty((block, el), dtype=torch.uint8, pin_memory=True) for _ in range(nbuf)] if cuda else None
    events = [None] * nbuf
    pool = ThreadPoolExecutor(max_workers=nbuf)
    pending = [None] * nbuf

    def drain(i, s, B):
        events[i].synchronize()
        np.copyto(out[s:s + B], pins[i][:B].numpy())

    k = 0
    for s in range(0, N, block):
        xb = x[s:s + block]
        B = xb.shape[0]
        if dtype == "float":
            vec = xb.float().contiguous().view(torch.uint8).view(B, D * 4)
        elif dtype == "uint8":
            vec = xb.to(torch.uint8)
        else:
            vec = xb.to(torch.int8).view(torch.uint8)
        d32 = deg[s:s + B].to(torch.int32).contiguous().view(torch.uint8).view(B, 4)
        a32 = adj[s:s + B].to(torch.int32).contiguous().view(torch.uint8).view(B, 4 * R)
        ent = torch.cat([vec, d32, a32], 1)
        if not cuda:
            out[s:s + B] = ent.numpy()
            continue
        i = k % nbuf
        if pending[i] is not None:
            pending[i].result()
        pins[i][:B].copy_(ent, non_blocking=True)
        events[i] = torch.cuda.Event()
        events[i].record()
        pending[i] = pool.submit(drain, i, s, B)
        k += 1
    for f in pending:
        if f is not None:
            f.result()
    pool.shutdown()
    return out


def graph_coverage(x: torch.Tensor, deg: torch.Tensor, adj: torch.Tensor, k: int = 10, sample: int = 2000, seed: int = 1) -> float:
    """Diagnostic: the share of the true k nearest neighbours of `sample` random points that their adjacency lists contain."""
    g = synth._gen(seed, x.device)
    pick = torch.randint(0, x.shape[0], (sample,), generator=g, device=x.device)
    ti, _ = synth.knn(x, x[pick].float(), k + 1, row_block=1024)
    hit = 0
    for r in range(sample):
        p = int(pick[r])
        true = [t for t in ti[r].tolist() if t != p][:k]
        row = set(adj[p, : int(deg[p])].tolist())
        hit += sum(1 for t in true if t in row)
    return hit / (sample * k)


def make_index_large(N: int, D: int, dtype: str, R: int, m: int, Q: int, K: int = 10, n_clusters: int = 1024,
                     seed: int = synth.SEED, device="cuda", pq_iters: int = 6, log=None, diag: bool = False, noise: float = 2.0, **graph_kw):
    """A structured index of N >= a few million points with brute-force ground truth for Q queries.
    Returns (Index, queries np [Q, D], gt_ids np u32 [Q, K], gt_dists np f32 [Q, K])."""
    t0 = time.time()
    big = N > 20_000_000
    x = synth.make_vectors(N, D, dtype, n_clusters=n_clusters, seed=seed, device=device, noise=noise,
                           out_dtype=(torch.bfloat16 if dtype != "float" and str(device).startswith("cuda") else None))
    if log:
        log(f"[build] {N} vectors generated in {time.time() - t0:.1f}s")
    xh = [x]                                                     # (a sliced build swaps the bf16 tensor for its 8-bit original half-way: see build_graph_large)
    del x
    deg, adj = build_graph_large(xh, R, seed=seed, log=log, idx_dtype=(torch.int32 if big else torch.int64),
                                 narrow=(dtype if dtype in ("uint8", "int8") else None), **graph_kw)
    x = xh[0]
    del xh
    medoid = synth.medoid_of(x)
    if diag and log:
        t1 = time.time()
        log(f"[build] the adjacency lists hold {100 * graph_coverage(x, deg, adj):.1f}% of the true 10-NN of 2000 sampled points ({time.time() - t1:.1f}s)")
    t1 = time.time()
    pivots, centroid, off, codes = synth.train_pq(x, m, iters=pq_iters, seed=seed)
    if log:
        log(f"[build] PQ: {m} chunks trained and {N} points encoded in {time.time() - t1:.1f}s")
    q = synth.make_queries(x, Q, dtype, seed=seed)
    t1 = time.time()
    gt_i, gt_d = synth.knn_big(x, q, K) if big else synth.knn(x, q.to(x.dtype), K, row_block=2048)
    if log:
        log(f"[build] brute-force ground truth for {Q} queries in {time.time() - t1:.1f}s")
    t1 = time.time()
    graph = pack_graph_device(x, dtype, deg, adj)
    del x, deg, adj
    if str(device).startswith("cuda"):
        torch.cuda.empty_cache()
    if log:
        log(f"[build] graph entries assembled on the host in {time.time() - t1:.1f}s")
    ix = Index(dtype=dtype, N=N, D=D, R=R, m=m, medoid=medoid, graph=graph, codes=codes.cpu().numpy(),
               pivots=pivots.cpu().numpy().astype(np.float32), centroid=centroid.cpu().numpy().astype(np.float32), chunk_off=off)
    if log:
        log(f"[build] index of {N} points built in {time.time() - t0:.1f}s")
    return ix, synth.to_numpy(q.float(), dtype), gt_i.cpu().numpy().astype(np.uint32), gt_d.cpu().numpy().astype(np.float32)
