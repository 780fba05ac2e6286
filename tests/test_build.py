"""Index construction tooling: partitioned candidate search + robust prune + long-range links."""
import numpy as np
import torch

from bang_amd import index_build as build, synth
from bang_amd.formats import Index, pack_graph


def _ref_prune(xp, cand, cand_d, keep, alpha):
    """Straightforward per-point robust prune (DiskANN's RobustPrune over a fixed candidate list, squared distances)."""
    alive = [np.isfinite(d) for d in cand_d]
    out = []
    for k, c in enumerate(cand):
        if not alive[k] or len(out) >= keep:
            continue
        out.append(k)
        for j in range(len(cand)):
            d_kc = float(((xp[c] - xp[cand[j]]) ** 2).sum())
            if alpha * alpha * d_kc <= cand_d[j]:
                alive[j] = False
    return out


def test_robust_prune_matches_a_plain_implementation():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(400, 8, generator=g)
    ci, cd = synth.knn(x, x, 12, exclude_self=True)
    mask = build.robust_prune(x, ci, cd, keep=5, alpha=1.2, block=128)
    xn = x.numpy().astype(np.float64)
    for p in range(0, 400, 7):
        want = _ref_prune(xn, ci[p].tolist(), cd[p].tolist(), 5, 1.2)
        assert mask[p].nonzero().squeeze(1).tolist() == want, p
    assert int(mask.sum(1).max()) <= 5 and bool(mask[:, 0].all())          # the nearest candidate always survives


def test_candidate_search_finds_most_true_neighbours():
    x = synth.make_vectors(6000, 32, "uint8", n_clusters=24, seed=5)
    ci, cd = build.candidate_neighbours(x, 16, probes=6, seed=5)
    ti, _ = synth.knn(x, x, 16, exclude_self=True)
    hit = np.mean([len(set(a) & set(b)) / 16.0 for a, b in zip(ci.tolist(), ti.tolist())])
    assert hit >= 0.9
    assert (np.diff(cd.numpy(), axis=1) >= 0).all() and (ci != torch.arange(6000)[:, None]).all()


def test_large_builder_makes_a_searchable_index():
    """The Vamana-style graph is navigable: the oracle reaches a recall comparable to the exact-kNN builder at the same L."""
    from oracle import oracle as O
    N, D, R, m, Q = 8000, 32, 32, 8, 64
    x = synth.make_vectors(N, D, "uint8", n_clusters=32, seed=9)
    deg, adj = build.build_graph_large(x, R, seed=9, K=24, probes=6)
    a = adj.numpy()
    d = deg.numpy()
    assert d.max() <= R and d.min() >= R // 2
    for i in range(0, N, 97):
        row = a[i, : d[i]]
        assert (np.diff(row) > 0).all() and i not in row and (a[i, d[i]:] == 0).all()
    piv, cen, off, codes = synth.train_pq(x, m, iters=4, seed=9)
    q = synth.make_queries(x, Q, "uint8", seed=9)
    gt_i, gt_d = synth.knn(x, q, 10)
    medoid = int(synth._sq_norms(x - x.mean(dim=0)).argmin())
    ix = Index("uint8", N, D, R, m, medoid, pack_graph(synth.to_numpy(x, "uint8"), d.astype(np.uint32), a.astype(np.uint32)),
               codes.numpy(), piv.numpy().astype(np.float32), cen.numpy().astype(np.float32), off)
    ids, _ = O.Oracle(ix).search(synth.to_numpy(q, "uint8"), 10, 60)
    rec = O.recall(gt_i.numpy().astype(np.uint32), gt_d.numpy().astype(np.float32), ids, 10)
    deg2, adj2 = synth.build_graph(x, R, seed=9)
    ix2 = Index("uint8", N, D, R, m, medoid, pack_graph(synth.to_numpy(x, "uint8"), deg2.numpy().astype(np.uint32),
                                                        adj2.numpy().astype(np.uint32)),
                codes.numpy(), piv.numpy().astype(np.float32), cen.numpy().astype(np.float32), off)
    ids2, _ = O.Oracle(ix2).search(synth.to_numpy(q, "uint8"), 10, 60)
    rec2 = O.recall(gt_i.numpy().astype(np.uint32), gt_d.numpy().astype(np.float32), ids2, 10)
    assert rec >= 80.0 and rec >= rec2 - 5.0, (rec, rec2)


def test_knn_big_equals_exact_knn():
    """synth.knn_big (the ground truth of the 1e8-2.5e8-point legs: shortlists per 65 536-column tile, exact re-evaluation) returns
    what the plain exact search returns -- ids and distances -- on 8-bit-valued data spanning several tiles, ties included."""
    import torch
    from bang_amd import synth
    g = torch.Generator().manual_seed(3)
    base = torch.randint(0, 256, (300_000, 64), generator=g).float()
    base[1000:1040] = base[5]                               # exact duplicates: ties are ordered by id in both
    q = base[torch.randint(0, base.shape[0], (40,), generator=g)] + torch.randint(-2, 3, (40, 64), generator=g).float()
    i1, d1 = synth.knn(base, q, 10, row_block=40)
    i2, d2 = synth.knn_big(base, q, 10)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)


def test_sliced_build_equals_the_one_shot_build():
    """build_graph_large with its per-point tables held for one slice of the points at a time (what lets a 3e8-point index fit HBM)
    returns the graph of the one-shot build, with and without the swap of the bf16 vectors for their 8-bit original."""
    x = synth.make_vectors(30_000, 32, "uint8", n_clusters=24, seed=5)
    deg1, adj1 = build.build_graph_large(x, 32, seed=5, K=24, probes=4, n_random=8, cell=512)
    deg3, adj3 = build.build_graph_large(x, 32, seed=5, K=24, probes=4, n_random=8, cell=512, slices=3)
    assert torch.equal(deg1, deg3) and torch.equal(adj1, adj3)
    xh = [x.to(torch.bfloat16)]
    deg4, adj4 = build.build_graph_large(xh, 32, seed=5, K=24, probes=4, n_random=8, cell=512, slices=2, narrow="uint8")
    assert xh[0].dtype == torch.uint8 and torch.equal(xh[0].float(), x)
    deg5, adj5 = build.build_graph_large(x.to(torch.bfloat16), 32, seed=5, K=24, probes=4, n_random=8, cell=512)
    assert torch.equal(deg4, deg5) and torch.equal(adj4, adj5)
