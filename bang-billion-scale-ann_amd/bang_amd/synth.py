"""Synthetic index builder (Tier A "structured" data of SURVEY.md 8(d)).

The reference ships no dataset (sift10kfiles.tar.gz is absent) and relies on external DiskANN
tools (``build_disk_index``, ``compute_groundtruth``; BANG_Base/ReadMe.pdf p.1-2) to produce the
files ``bang_load`` reads.  This module produces the same files from a seed:

* base vectors  = low-intrinsic-dimension Gaussian mixture (16-d latent mapped to D dims), clipped/rounded
                  for uint8 / int8;
* graph         = R/2 exact nearest neighbours + R/2 uniformly random long links per node,
                  de-duplicated, adjacency sorted ascending (bang_preprocess.py:102-104), ragged
                  degree allowed;
* medoid        = point nearest the dataset mean;
* PQ            = DiskANN-style: centroid = mean, D split into m chunks of near-equal size,
                  256-means per chunk on centred data, codes = nearest pivot;
* queries       = base points + Gaussian noise; ground truth by brute force.

Everything runs through torch so the same code builds 10K-point fixtures on the CPU and the
1M-point benchmark index on the GPU in seconds.  (Index *construction* is outside the search
hot path -- SURVEY 8 f-3.)
"""
from __future__ import annotations

import numpy as np
import torch

from .formats import Index, NP_DTYPE, pack_graph

SEED = 20240711


def _gen(seed: int, device) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def make_vectors(N: int, D: int, dtype: str, n_clusters: int = 256, d_lat: int = 16, spread: float = 3.0,
                 noise: float = 2.0, seed: int = SEED, device="cpu", out_dtype=None) -> torch.Tensor:
    """float32 tensor [N, D] already rounded/clipped to the value range of ``dtype``.

    Low intrinsic dimension, like real descriptor data: a ``d_lat``-dimensional Gaussian-mixture latent
    (``n_clusters`` centres with std ``spread``, unit within-cluster std) is mapped to D dims by a random
    matrix with unit columns, scaled to a per-dimension std of 36 around 128 and perturbed by N(0, noise).
    (An isotropic D-dimensional mixture makes every in-cluster point equidistant -- neither the graph nor
    PQ can rank neighbours there and recall needs L > 400.)"""
    g = _gen(seed, device)
    A = torch.randn(d_lat, D, generator=g, device=device)
    A = A / A.norm(dim=0, keepdim=True)
    centres = torch.randn(n_clusters, d_lat, generator=g, device=device) * spread
    if out_dtype is not None and N > (1 << 24):
        # large N: generated block by block into a tensor of out_dtype (bfloat16 holds 8-bit values exactly) -- the [N, D] float32
        # intermediate of the one-shot path is 51 GB at N = 1e8.  Same distribution, its own random stream; the scale comes from
        # the first block.
        out = torch.empty((N, D), dtype=out_dtype, device=device)
        scale = None
        for a in range(0, N, 1 << 22):
            n = min(1 << 22, N - a)
            assign = torch.randint(0, n_clusters, (n,), generator=g, device=device)
            xb = (centres[assign] + torch.randn(n, d_lat, generator=g, device=device)) @ A
            if scale is None:
                scale = 36.0 / float(xb.std())
            xb = xb * scale + 128.0 + torch.randn(n, D, generator=g, device=device) * noise
            if dtype == "uint8":
                xb = xb.round().clamp_(0, 255)
            elif dtype == "int8":
                xb = (xb - 128.0).round().clamp_(-128, 127)
            else:
                xb = (xb - 128.0) / 128.0
            out[a:a + n] = xb.to(out_dtype)
        return out
    assign = torch.randint(0, n_clusters, (N,), generator=g, device=device)
    z = centres[assign] + torch.randn(N, d_lat, generator=g, device=device)
    x = z @ A
    x = x * (36.0 / float(x.std())) + 128.0 + torch.randn(N, D, generator=g, device=device) * noise
    if dtype == "uint8":
        x = x.round().clamp_(0, 255)
    elif dtype == "int8":
        x = (x - 128.0).round().clamp_(-128, 127)
    else:
        x = (x - 128.0) / 128.0
    return x if out_dtype is None else x.to(out_dtype)


def medoid_of(x: torch.Tensor) -> int:
    """The point nearest the dataset mean (block-wise: x may be a 25 GB bfloat16 tensor)."""
    N = x.shape[0]
    mean = torch.zeros(x.shape[1], dtype=torch.float64, device=x.device)
    for a in range(0, N, 1 << 22):
        mean += x[a:a + (1 << 22)].double().sum(0)
    mean = (mean / N).float()
    best, arg = float("inf"), 0
    for a in range(0, N, 1 << 22):
        d = _sq_norms(x[a:a + (1 << 22)].float() - mean)
        v, i = torch.min(d, 0)
        if float(v) < best:
            best, arg = float(v), a + int(i)
    return arg


def _sq_norms(x: torch.Tensor) -> torch.Tensor:
    return (x * x).sum(dim=1)


def knn(base: torch.Tensor, queries: torch.Tensor, k: int, exclude_self: bool = False,
        row_block: int = 4096, col_block: int = 262144):
    """Exact k nearest neighbours (squared L2) of every query row among ``base`` rows.
    Returns (ids int64 [Q,k], dists float32 [Q,k]) sorted ascending by (dist, id)."""
    Nb = base.shape[0]
    low = base.dtype != torch.float32                     # 8-bit data held as bfloat16: exact values, f32 arithmetic per block
    bn = torch.cat([_sq_norms(base[a:a + (1 << 22)].float()) for a in range(0, Nb, 1 << 22)]) if low else _sq_norms(base)
    out_i, out_d = [], []
    for r0 in range(0, queries.shape[0], row_block):
        q = queries[r0:r0 + row_block].float()
        qn = _sq_norms(q)
        cand_d, cand_i = [], []
        for c0 in range(0, Nb, col_block):
            b = base[c0:c0 + col_block].float()
            d = qn[:, None] + bn[None, c0:c0 + col_block] - 2.0 * (q @ b.T)
            if exclude_self:
                rows = torch.arange(r0, r0 + q.shape[0], device=base.device)
                inside = (rows >= c0) & (rows < c0 + b.shape[0])
                d[inside.nonzero().squeeze(1), rows[inside] - c0] = float("inf")
            kk = min(k, b.shape[0])
            dd, ii = torch.topk(d, kk, dim=1, largest=False)
            cand_d.append(dd)
            cand_i.append(ii + c0)
        cd = torch.cat(cand_d, dim=1)
        ci = torch.cat(cand_i, dim=1)
        # exact re-evaluation of the shortlisted candidates (removes matmul round-off), then a
        # lexicographic (dist, id) order so ties are deterministic
        diff = q[:, None, :] - base[ci].float()
        cd = (diff * diff).sum(dim=2)
        order = torch.argsort(ci, dim=1, stable=True)
        cd, ci = torch.gather(cd, 1, order), torch.gather(ci, 1, order)
        order = torch.argsort(cd, dim=1, stable=True)[:, :k]
        out_d.append(torch.gather(cd, 1, order))
        out_i.append(torch.gather(ci, 1, order))
    return torch.cat(out_i), torch.cat(out_d)


def knn_big(base: torch.Tensor, queries: torch.Tensor, k: int, col_block: int = 65536, group: int = 256):
    """Exact k nearest neighbours of a few thousand queries among up to 1e8+ base rows (ground truth of the 100 M index) without a
    top-k over every distance tile: a tile [Q, col_block] is viewed as [Q, col_block / group, group]; the k groups with the smallest
    minima contain the tile's k nearest (a group holding one of them has a minimum <= the k-th smallest value, and at most k groups
    do), so one min-reduction + a top-k over 256 group minima + a top-k over k x group gathered values replace the big selection.
    The tile distances are float32 |b|^2 - 2 q.b of magnitude ~1e7 for 8-bit data, i.e. good to a few units: a true neighbour within
    rounding of a tile's k-th candidate could miss a shortlist of exactly k -- so 2k groups and 2k ids per tile are kept.
    The shortlisted ids are re-evaluated exactly and ordered by (distance, id) like knn()."""
    Nb, D = base.shape
    dev = base.device
    q = queries.float()
    Q = q.shape[0]
    qn = _sq_norms(q)
    ci_all, ar = [], torch.arange(group, device=dev)
    for c0 in range(0, Nb, col_block):
        b = base[c0:c0 + col_block].float()
        nb = b.shape[0]
        d = torch.addmm(_sq_norms(b)[None, :], q, b.T, alpha=-2.0)              # (+ |q|^2: constant per row, irrelevant for the order)
        if nb % group:
            d = torch.cat([d, torch.full((Q, group - nb % group), float("inf"), device=dev)], 1)
        ng = d.shape[1] // group
        dv = d.view(Q, ng, group)
        gsel = torch.topk(dv.amin(dim=2), min(2 * k, ng), dim=1, largest=False).indices      # [Q, 2k] groups
        sub = torch.gather(dv, 1, gsel[:, :, None].expand(-1, -1, group)).reshape(Q, -1)      # [Q, 2k * group]
        ii = torch.topk(sub, min(2 * k, sub.shape[1]), dim=1, largest=False).indices
        ids = torch.gather(gsel, 1, ii // group) * group + ii % group + c0
        ci_all.append(ids.clamp_(max=Nb - 1))
    ci = torch.cat(ci_all, 1)
    out_i, out_d = [], []
    rb = max(8, min(256, (1 << 26) // max(1, ci.shape[1] * D)))                   # rows per block: the gathered candidates stay below ~256 MB
    for r0 in range(0, Q, rb):                                                    # exact re-evaluation of the shortlist
        c = ci[r0:r0 + rb]
        diff = q[r0:r0 + rb, None, :] - base[c].float()
        cd = (diff * diff).sum(2)
        order = torch.argsort(c, dim=1, stable=True)
        cd, c = torch.gather(cd, 1, order), torch.gather(c, 1, order)
        dup = torch.zeros_like(c, dtype=torch.bool)
        dup[:, 1:] = c[:, 1:] == c[:, :-1]
        cd = torch.where(dup, torch.full_like(cd, float("inf")), cd)
        order = torch.argsort(cd, dim=1, stable=True)[:, :k]
        out_d.append(torch.gather(cd, 1, order))
        out_i.append(torch.gather(c, 1, order))
    return torch.cat(out_i), torch.cat(out_d)


def build_graph(x: torch.Tensor, R: int, seed: int = SEED, n_knn: int | None = None):
    """kNN + random-long-link graph.  Returns (degrees int64 [N], adjacency int64 [N,R]) with each
    row's first ``degree`` entries distinct, != self and sorted ascending; the tail is 0."""
    N = x.shape[0]
    dev = x.device
    n_knn = min(R // 2 if n_knn is None else n_knn, N - 1)
    nn_ids, _ = knn(x, x, n_knn, exclude_self=True)
    g = _gen(seed + 1, dev)
    rnd = torch.randint(0, N, (N, R - n_knn), generator=g, device=dev)
    adj = torch.cat([nn_ids, rnd], dim=1)
    adj, _ = torch.sort(adj, dim=1)
    self_id = torch.arange(N, device=dev)[:, None]
    keep = torch.ones_like(adj, dtype=torch.bool)
    keep[:, 1:] = adj[:, 1:] != adj[:, :-1]
    keep &= adj != self_id
    deg = keep.sum(dim=1)
    pos = torch.cumsum(keep.to(torch.int64), dim=1) - 1
    out = torch.zeros_like(adj)
    rows = torch.arange(N, device=dev)[:, None].expand_as(adj)
    out[rows[keep], pos[keep]] = adj[keep]
    return deg, out


def chunk_offsets(D: int, m: int) -> np.ndarray:
    """DiskANN's split: the first D % m chunks get ceil(D/m) dims, the rest floor(D/m)."""
    lo, rem = divmod(D, m)
    sizes = [lo + 1 if c < rem else lo for c in range(m)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)


def train_pq(x: torch.Tensor, m: int, iters: int = 8, sample: int = 65536, seed: int = SEED):
    """Returns (pivots [256,D] f32, centroid [D] f32, chunk_off [m+1] u32, codes [N,m] u8) as torch/np."""
    N, D = x.shape
    dev = x.device
    g = _gen(seed + 2, dev)
    if x.dtype == torch.float32 and N <= (1 << 24):
        centroid = x.mean(dim=0)
    else:                                                  # block-wise (a 25 GB bfloat16 tensor must not be copied to float32 whole)
        acc = torch.zeros(D, dtype=torch.float64, device=dev)
        for a in range(0, N, 1 << 22):
            acc += x[a:a + (1 << 22)].double().sum(0)
        centroid = (acc / N).float()
    off = chunk_offsets(D, m)
    pivots = torch.zeros(256, D, device=dev)
    codes = torch.empty(N, m, dtype=torch.uint8, device=dev)
    sidx = torch.randperm(N, generator=g, device=dev)[: min(sample, N)]
    xs = x[sidx].float() - centroid
    cens = []
    for c in range(m):
        a, b = int(off[c]), int(off[c + 1])
        sub = xs[:, a:b]
        if sub.shape[0] >= 256:
            init = torch.randperm(sub.shape[0], generator=g, device=dev)[:256]
        else:
            init = torch.randint(0, sub.shape[0], (256,), generator=g, device=dev)
        cen = sub[init].clone()
        for _ in range(iters):
            d = torch.cdist(sub, cen)
            lab = d.argmin(dim=1)
            sums = torch.zeros_like(cen).index_add_(0, lab, sub)
            cnt = torch.zeros(256, device=dev).index_add_(0, lab, torch.ones_like(lab, dtype=torch.float32))
            nz = cnt > 0
            cen[nz] = sums[nz] / cnt[nz, None]
        pivots[:, a:b] = cen
        cens.append(cen)
    # encode: chunks of equal width are handled together ([rows, chunks, 256] distances from one batched product)
    widths = sorted(set(int(off[c + 1] - off[c]) for c in range(m)))
    groups = []
    for w in widths:
        cs = [c for c in range(m) if int(off[c + 1] - off[c]) == w]
        cols = torch.tensor([int(off[c]) + j for c in cs for j in range(w)], device=dev)
        cg = torch.stack([cens[c] for c in cs])                                     # [nc, 256, w]
        groups.append((torch.tensor(cs, device=dev), cols, cg, (cg * cg).sum(2), w))
    step = 1 << 20 if N <= (1 << 24) else 1 << 16
    for r0 in range(0, N, step):
        full = x[r0:r0 + step].float() - centroid
        B = full.shape[0]
        for cs_t, cols, cg, cn, w in groups:
            sub = full[:, cols].view(B, cs_t.shape[0], w).transpose(0, 1)           # [nc, B, w]
            d = cn[:, None, :] - 2.0 * torch.bmm(sub, cg.transpose(1, 2))           # [nc, B, 256]  (+ |x|^2: constant per row)
            codes[r0:r0 + B, cs_t] = d.argmin(dim=2).transpose(0, 1).to(torch.uint8)
    return pivots, centroid, off, codes


def make_queries(x: torch.Tensor, Q: int, dtype: str, noise: float = 4.5, seed: int = SEED) -> torch.Tensor:
    g = _gen(seed + 3, x.device)
    pick = torch.randint(0, x.shape[0], (Q,), generator=g, device=x.device)
    scale = noise if dtype != "float" else noise / 128.0
    q = x[pick].float() + torch.randn(Q, x.shape[1], generator=g, device=x.device) * scale
    if dtype == "uint8":
        q = q.round().clamp_(0, 255)
    elif dtype == "int8":
        q = q.round().clamp_(-128, 127)
    return q


def to_numpy(x: torch.Tensor, dtype: str) -> np.ndarray:
    return x.detach().cpu().numpy().astype(NP_DTYPE[dtype])


def make_index(N: int, D: int, dtype: str, R: int, m: int, Q: int, K: int = 10, n_clusters: int = 256,
               seed: int = SEED, device="cpu", pq_iters: int = 8):
    """Build a complete structured test case.  Returns (Index, queries np [Q,D], gt_ids np u32 [Q,K],
    gt_dists np f32 [Q,K])."""
    x = make_vectors(N, D, dtype, n_clusters=n_clusters, seed=seed, device=device)
    deg, adj = build_graph(x, R, seed=seed)
    medoid = int(_sq_norms(x - x.mean(dim=0)).argmin())
    pivots, centroid, off, codes = train_pq(x, m, iters=pq_iters, seed=seed)
    q = make_queries(x, Q, dtype, seed=seed)
    gt_i, gt_d = knn(x, q, K)
    vec_np = to_numpy(x, dtype)
    graph = pack_graph(vec_np, deg.cpu().numpy().astype(np.uint32), adj.cpu().numpy().astype(np.uint32))
    ix = Index(dtype=dtype, N=N, D=D, R=R, m=m, medoid=medoid, graph=graph,
               codes=codes.cpu().numpy(), pivots=pivots.cpu().numpy().astype(np.float32),
               centroid=centroid.cpu().numpy().astype(np.float32), chunk_off=off)
    return ix, to_numpy(q, dtype), gt_i.cpu().numpy().astype(np.uint32), gt_d.cpu().numpy().astype(np.float32)
