"""BASELINE.json configs[1] at FULL size on the GPU (N = 1 M, uint8, D = 128, R = 64, m = 32, Q = 10 000): the whole batch
through size-independent properties, across every graph form, and -- the oracle's OpenMP loop answers 10 000 queries in about a
second on the GPU box's cores -- every query bit for bit against the oracle.  The index is built on the GPU with torch
(plumbing only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sift1m_like():
    import torch
    from bang_amd import synth
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return synth.make_index(1_000_000, 128, "uint8", 64, 32, 10_000, K=10, n_clusters=256, device="cuda")


def _search(ix, q, graph, L=70, **opts):
    import bang_amd
    with bang_amd.Engine(ix.dtype, graph=graph, **opts) as e:
        e.load_index(ix)
        e.set_searchparams(10, L)
        e.alloc(q.shape[0])
        e.init(q.shape[0])
        ids, dists = e.query(q)
        st = e.stats()
        e.free()
        e.unload()
    return ids, dists, st


def test_full_size_properties_and_sample_parity(libbang, sift1m_like):
    from oracle import oracle as O
    ix, q, gt_i, gt_d = sift1m_like
    ids_h, dists_h, st_h = _search(ix, q, 0, pull=0)                      # host graph served by the C++ walker threads
    ids_p, dists_p, st_p = _search(ix, q, 0)                              # host graph, default: the kernel pulls the rows over PCIe
    assert st_h["graph_pull"] == 0 and st_p["graph_pull"] == 1 and st_p["pulled_bytes"] == 256 * (st_p["candidates"] - 10_000)
    assert np.array_equal(ids_h, ids_p) and np.array_equal(dists_h.view(np.uint32), dists_p.view(np.uint32))
    assert st_h["dist_evals"] == st_p["dist_evals"] and st_h["fetched"] == st_p["fetched"]
    ids_d, dists_d, st_d = _search(ix, q, 1)
    ids_l, dists_l, st_l = _search(ix, q, 0, persistent=0, vectors=0)   # host graph, a launch per iteration and lane, vectors shipped by the walker
    ids_v, dists_v, st_v = _search(ix, q, 0, vectors=0)                 # persistent kernel, vectors shipped by the walker
    assert np.array_equal(ids_h, ids_v) and np.array_equal(dists_h.view(np.uint32), dists_v.view(np.uint32))
    # 1. neither graph placement nor the form of the host loop changes a single bit of the result
    assert np.array_equal(ids_h, ids_d) and np.array_equal(dists_h.view(np.uint32), dists_d.view(np.uint32))
    assert np.array_equal(ids_h, ids_l) and np.array_equal(dists_h.view(np.uint32), dists_l.view(np.uint32))
    assert st_h["dist_evals"] == st_d["dist_evals"] == st_l["dist_evals"] and st_h["candidates"] == st_d["candidates"] == st_l["candidates"]
    assert st_h["fetched"] == st_l["fetched"]
    # 2. size-independent properties over the WHOLE batch
    assert ids_h.shape == (10_000, 10) and (ids_h < ix.N).all()
    srt = np.sort(ids_h, axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                                  # 10 distinct neighbours per query
    assert (np.diff(dists_h, axis=0) >= 0).all()                              # ascending exact distances ([rank][Q])
    vec = ix.vectors()
    for a in range(0, 10_000, 2000):                                          # returned distance == exact squared L2 (integers: exact)
        blk = vec[ids_h[a:a + 2000].astype(np.int64)].astype(np.int32) - q[a:a + 2000, None, :].astype(np.int32)
        assert np.array_equal((blk * blk).sum(axis=2).T.astype(np.float32), dists_h[:, a:a + 2000])
    assert O.recall(gt_i, gt_d, ids_h, 10) >= 88.0
    # 3. a sample against the oracle, bit for bit
    sel = np.arange(0, 10_000, 79)[:128]
    ids_o, dists_o = O.Oracle(ix).search(q[sel], 10, 70)
    assert np.array_equal(ids_h[sel], ids_o)
    assert np.array_equal(dists_h[:, sel].view(np.uint32), dists_o.view(np.uint32))
    # 4. iteration accounting: nobody exceeds the cap (bang_search.cu:950), and the worklist length bounds the candidates
    assert st_h["iterations"] <= 70 + 49 and st_h["candidates"] <= 10_000 * (70 + 50)


@pytest.mark.parametrize("L", [37, 152])
def test_full_size_every_query_matches_oracle(libbang, sift1m_like, L):
    """All 10 000 queries of the full-size batch, bit for bit against the oracle (ids, distances, per-query counters), with the
    graph in HBM and in host RAM (rows pulled by the kernel): 4 096 waves handing out queries dynamically, worklists that fill
    early (L = 37: more entering survivors than slots in the first iterations) and late (L = 152)."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = sift1m_like
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, L, with_stats=True)
    for opts in (dict(graph=1), dict(graph=0, pull=1)):
        with bang_amd.Engine(ix.dtype, **opts) as e:
            e.load_index(ix)
            e.set_searchparams(10, L)
            e.alloc(q.shape[0])
            e.init(q.shape[0])
            ids, dists = e.query(q)
            cnt = e.query_counters(q.shape[0])
            e.free()
            e.unload()
        assert np.array_equal(ids, ids_o), opts
        assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), opts
        assert np.array_equal(cnt, st_o), opts
