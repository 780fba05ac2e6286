"""Launch policies of the search kernel never change results (`spec_rows` moves the request for the PQ code rows in front of the filter,
bang_search.cu:1140-1165 / :1201-1241: distances of ids the filter drops are computed and discarded; the filter summary only drops requests
whose answer is known).

The fixtures' own batches (40-64 queries) are far below a full chip, where both policies resolve to "off": here the batch is LARGE
(more queries than wave slots, so the launch is full and queries are handed out from the queue) and the policies are forced both ways,
on the long-row instances (m = 70 / 74: 12 waves, cooperative row fetch) and a short-row one (m = 32: 16 waves, per-lane loads)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_big = {}


def _big_batch(request, fixture, Q):
    """The fixture's index with a batch of Q queries (its own queries first, then random points of the same type) + the oracle's answer."""
    if fixture not in _big:
        from oracle import oracle as O
        ix, q, _, _ = request.getfixturevalue(fixture)
        rng = np.random.default_rng(99)
        if q.dtype == np.float32:
            more = (q[rng.integers(0, q.shape[0], Q - q.shape[0])] + rng.normal(0, 0.05, (Q - q.shape[0], q.shape[1]))).astype(np.float32)
        else:
            info = np.iinfo(q.dtype)
            more = rng.integers(info.min, info.max + 1, (Q - q.shape[0], q.shape[1])).astype(q.dtype)
        qq = np.ascontiguousarray(np.concatenate([q, more]))
        ids_o, dists_o, st_o = O.Oracle(ix).search(qq, 10, 64, with_stats=True)
        _big[fixture] = (ix, qq, ids_o, dists_o, st_o)
    return _big[fixture]


@pytest.mark.parametrize("summ_iters,max_wgs,spec_rows,prio", [("-1", "0", "1", "1"), ("3", "0", "2", "0"), ("3", "96", "1", "1"), ("-1", "96", "2", "0"), ("0", "0", "0", "")])
@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("fixture", ["small_u8", "small_deep", "small_f32"])
def test_full_launch_policies_match_oracle(request, libbang, monkeypatch, fixture, graph, summ_iters, max_wgs, spec_rows, prio):
    import bang_amd
    ix, qq, ids_o, dists_o, st_o = _big_batch(request, fixture, 4300)        # > 256 CUs x 16 waves: queries handed out from the queue in every instance -- as two equal
    # rounds of 9 waves per CU (bang_search_geometry) on the whole chip, as four to five rounds of every wave that fits on 96 workgroups
    monkeypatch.setenv("BANG_SUMM_ITERS", summ_iters)
    monkeypatch.setenv("BANG_SEARCH_MAX_WGS", max_wgs)
    monkeypatch.setenv("BANG_SPEC_ROWS", spec_rows)          # (code rows requested with the filter probes: the 70-chunk fixture has the instance)
    if prio:
        monkeypatch.setenv("BANG_SEARCH_PRIO", prio)         # (raised wave priority around the request-issuing stretches: forced on / off; "" = the launch policy)
    with bang_amd.Engine(ix.dtype, graph=graph, search=1) as e:
        e.load_index(ix)
        e.set_searchparams(10, 64)
        e.alloc(qq.shape[0])
        e.init(qq.shape[0])
        ids, dists = e.query(qq)
        st = e.stats()
        ctr = e.query_counters(qq.shape[0])
        e.free(); e.unload()
    assert st["search_kernel"] == 1
    assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
    assert np.array_equal(ctr, st_o.astype(np.int64))                         # iterations, expansions, evaluations, ids offered: per query


def test_soak_slice(libbang):
    """A 60-second slice of tools/soak_random.py: random index shapes and search parameters, three loop forms, launch policies drawn at random."""
    sys.path.insert(0, ROOT)
    from tools import soak_random
    lines = []
    done, batches, bad = soak_random.run(n_cases=400, seed=int(os.environ.get("BANG_SOAK_SEED", "20261003")), budget_s=60.0, log=lines.append)
    assert bad == 0, "\n".join(l for l in lines if "MISMATCH" in l)
    assert done >= 5 and batches >= 30


@pytest.mark.parametrize("L", [10, 152])
@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("fixture", ["small_u8", "small_i8", "small_f32", "small_deep"])
def test_fused_rerank_matches_oracle_and_the_rerank_launch(request, libbang, fixture, graph, L):
    """K6 + K7 inside the search launch (self-paced form: compute_L2Dist<T> bang_search.cu:1254-1299 -- one template over u8 / i8 / float --,
    compute_NearestNeighbours :1312-1368, by the wave that finishes the query) against the oracle and against the separate re-rank launch: ids,
    distance bits, [rank][Q] layout.  8-bit vectors: D / 16 lanes per candidate, exact integers; float vectors (round 6): one lane per candidate
    runs the ascending fmaf chain."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    ids_o, dists_o = O.Oracle(ix).search(q, 10, L)
    out = {}
    for fuse in (1, 0):
        with bang_amd.Engine(ix.dtype, graph=graph, search=1, fuse_rerank=fuse) as e:
            e.load_index(ix)
            e.set_searchparams(10, L)
            e.alloc(q.shape[0])
            e.init(q.shape[0])
            out[fuse] = e.query(q)
            assert e.stats()["rerank_fused"] == fuse
            e.free(); e.unload()
    for fuse in (1, 0):
        assert np.array_equal(out[fuse][0], ids_o) and np.array_equal(out[fuse][1].view(np.uint32), dists_o.view(np.uint32)), fuse


def test_fused_rerank_float_vectors_odd_dimension_counts_and_short_logs(libbang):
    """Float vectors whose dimension count is no multiple of 16 (the last trip of a lane's chain loads fewer than four pieces) and one beyond
    64 dimensions with a partial second register of query elements; k > candidates (CANON 8 tail); both placements."""
    import bang_amd
    from bang_amd import synth
    from oracle import oracle as O
    for (N, D, R, m, seed) in ((2000, 40, 32, 16, 5), (1500, 72, 64, 16, 6)):
        ix, q, _, _ = synth.make_index(N, D, "float", R, m, 24, K=10, n_clusters=16, seed=seed, device="cpu", pq_iters=3)
        orc = O.Oracle(ix)
        for (k, L) in ((10, 30), (20, 21)):
            ids_o, dists_o = orc.search(q, k, L)
            for graph in (0, 1):
                with bang_amd.Engine(ix.dtype, graph=graph, search=1) as e:
                    e.load_index(ix)
                    e.set_searchparams(k, L)
                    e.alloc(q.shape[0])
                    e.init(q.shape[0])
                    ids, dists = e.query(q)
                    assert e.stats()["rerank_fused"] == 1 and e.stats()["search_kernel"] == 1, (D, e.stats())
                    e.free(); e.unload()
                assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), (D, k, L, graph)


def test_fused_rerank_short_logs_many_lanes_and_big_k(libbang, small_u8):
    """Fewer candidates than k (CANON 8: UINT64_MAX / BIG_DIST tail), k close to L, queries split over lanes (each lane's launch writes its
    own rows of the [Q][k] / [k][Q] result block), a batch that is handed out from the queue."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    orc = O.Oracle(ix)
    for (k, L, lanes, max_wgs) in ((30, 33, 1, "0"), (10, 48, 3, "0"), (64, 70, 2, "2"), (5, 5, 1, "1")):
        ids_o, dists_o = orc.search(q, k, L)
        os.environ["BANG_SEARCH_MAX_WGS"] = max_wgs
        try:
            with bang_amd.Engine(ix.dtype, graph=1, search=1, lanes=lanes) as e:
                e.load_index(ix)
                e.set_searchparams(k, L)
                e.alloc(q.shape[0])
                e.init(q.shape[0])
                ids, dists = e.query(q)
                assert e.stats()["rerank_fused"] == 1
                e.free(); e.unload()
        finally:
            os.environ.pop("BANG_SEARCH_MAX_WGS", None)
        assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), (k, L, lanes)


@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("direct", ["1", "0"])
def test_fused_rerank_results_straight_into_the_pinned_mirror(request, libbang, monkeypatch, small_u8, graph, direct):
    """With the re-rank fused, the kernel writes ids [Q][k], distances [rank][Q], the per-query iteration counts and its abort word straight
    into the pinned host mirror of the results (no copy behind the launch); BANG_RESULTS_DIRECT=0 keeps them in device memory and copies
    them back (compute_NearestNeighbours bang_search.cu:1312-1368 + the D2H at :997-999).  Same bits, same counters, twice in a row."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    monkeypatch.setenv("BANG_RESULTS_DIRECT", direct)
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, 48, with_stats=True)
    with bang_amd.Engine(ix.dtype, graph=graph, fuse_rerank=1) as e:
        e.load_index(ix)
        e.set_searchparams(10, 48)
        e.alloc(q.shape[0])
        for _ in range(2):
            e.init(q.shape[0])
            ids, dists = e.query(q)
            assert e.stats()["rerank_fused"] == 1
            assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
            assert np.array_equal(e.query_counters(q.shape[0]), st_o.astype(np.int64))
        e.free(); e.unload()

