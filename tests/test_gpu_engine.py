"""End-to-end parity on the GPU: libbang.so (HIP) vs the CPU oracle, through the C-ABI engine.

Bit-exact: returned neighbour ids (u64) and exact distances (f32) must be identical, for both
graph placements, every dtype and both PQ paths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_engine(ix, queries, k, L, **opts):
    import bang_amd
    distfn = opts.pop("distfn", bang_amd.DIST_L2)
    with bang_amd.Engine(ix.dtype, **opts) as e:
        e.load_index(ix)
        e.set_searchparams(k, L, distfn)
        e.alloc(queries.shape[0])
        e.init(queries.shape[0])
        ids, dists = e.query(queries)
        st = e.stats()
        # second run on the same allocation must reproduce the first (bang_init resets all state)
        e.init(queries.shape[0])
        ids2, dists2 = e.query(queries)
        assert np.array_equal(ids, ids2) and np.array_equal(dists, dists2)
        e.free()
        e.unload()
    return ids, dists, st


@pytest.mark.parametrize("fixture", ["small_f32", "small_u8", "small_deep", "small_i8"])
@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("L", [10, 37, 152])
def test_engine_matches_oracle(request, libbang, fixture, graph, L):
    from oracle import oracle as O
    ix, q, gt_i, gt_d = request.getfixturevalue(fixture)
    orc = O.Oracle(ix)
    ids_o, dists_o, st_o = orc.search(q, 10, L, with_stats=True)
    ids, dists, st = _run_engine(ix, q, 10, L, graph=graph)
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
    assert st["dist_evals"] == int(st_o[:, 2].sum())
    assert st["fetched"] == int(st_o[:, 3].sum())
    assert st["candidates"] == int(st_o[:, 1].sum())


@pytest.mark.parametrize("fixture", ["small_f32", "small_u8"])
def test_engine_lut_path_matches_oracle(request, libbang, fixture):
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    ids_o, dists_o = O.Oracle(ix).search(q, 10, 64)
    ids, dists, _ = _run_engine(ix, q, 10, 64, pq=1)
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))


@pytest.mark.parametrize("lanes", [1, 3, 8])
def test_engine_lanes_do_not_change_results(request, libbang, small_u8, lanes):
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    ids_o, dists_o = O.Oracle(ix).search(q, 10, 48)
    ids, dists, _ = _run_engine(ix, q, 10, 48, lanes=lanes, persistent=0)
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))


def test_engine_mips(request, libbang, small_f32):
    """MIPS mode: queries carry D-1 coordinates, the last one is implicitly 0 (bang_search.cu:1099-1113)."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_f32
    q1 = np.ascontiguousarray(q[:, :-1])
    ids_o, dists_o = O.Oracle(ix).search(q1, 10, 40, mips=True)
    ids, dists, _ = _run_engine(ix, q1, 10, 40, distfn=bang_amd.DIST_MIPS)
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))


def test_engine_recall_is_meaningful(request, libbang, small_u8):
    from oracle import oracle as O
    ix, q, gt_i, gt_d = small_u8
    ids, _, _ = _run_engine(ix, q, 10, 100)
    assert O.recall(gt_i, gt_d, ids, 10) >= 90.0


@pytest.mark.parametrize("opts", [dict(use_flag=0), dict(stage_zero_copy=0), dict(threads=1), dict(threads=3, lanes=2, persistent=0),
                                  dict(fp_batch=1, persistent=0), dict(front_wgs=7, lanes=3, persistent=0), dict(timing=1),
                                  dict(timing=1, persistent=0), dict(persistent=0), dict(persistent=1),
                                  dict(persistent=1, lanes=3, threads=2), dict(persistent=1, timing=1), dict(persistent=1, threads=1),
                                  dict(persistent=1, stage_zero_copy=1), dict(persistent=0, stage_zero_copy=1)])
@pytest.mark.parametrize("fixture", ["small_u8", "small_i8"])   # small_u8: PQ layout without a persistent instance; small_i8: with
def test_engine_host_loop_options_do_not_change_results(request, libbang, fixture, opts):
    """Every host-loop mechanism (persistent search kernel vs a launch per iteration, in-kernel completion flag vs runtime
    sync, BAR vs zero-copy vs copied adjacency rows, walker team size, vector-copy batching, CU share per lane, in-kernel
    timing) is a pure performance knob."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    ids_o, dists_o = O.Oracle(ix).search(q, 10, 33)
    with bang_amd.Engine(ix.dtype, graph=0, pull=0) as e:          # (pull = 0: the walker serves the graph; these are its knobs)
        for k, v in opts.items():
            if k == "use_flag":
                continue
            e.set_option(k, v)
        if "use_flag" in opts:
            import os
            os.environ["BANG_USE_FLAG"] = "0"
        try:
            e2 = bang_amd.Engine(ix.dtype, graph=0, pull=0) if "use_flag" in opts else e
            e2.load_index(ix)
            e2.set_searchparams(10, 33)
            e2.alloc(q.shape[0])
            for _ in range(2):
                e2.init(q.shape[0])
                ids, dists = e2.query(q)
                assert np.array_equal(ids, ids_o)
                assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
            if opts.get("timing"):
                st = e2.stats()
                if st["persistent"]:     # one launch: its duration contains the front phases of any one workgroup
                    assert 0 < st["front_ms"] <= st["front_busy_ms"] + 1e-9 and st["front_launches"] == 1
                else:                    # union of the launch intervals vs their sum
                    assert st["front_ms"] > 0 and 0 < st["front_busy_ms"] <= st["front_ms"] + 1e-9
            e2.free()
            e2.unload()
            if e2 is not e:
                e2.close()
        finally:
            os_env = __import__("os").environ
            os_env.pop("BANG_USE_FLAG", None)


@pytest.mark.parametrize("fixture", ["small_f32", "small_u8", "small_deep", "small_i8"])
@pytest.mark.parametrize("L", [10, 152])
@pytest.mark.parametrize("persistent", [0, 1])
@pytest.mark.parametrize("vectors", [0, 1])
def test_engine_host_loop_variants_match_oracle(request, libbang, fixture, L, persistent, vectors):
    """Host-graph mode, both loop forms: "persistent"=1 = ONE search kernel per batch whose workgroups are paced one by one
    by the walker threads; "persistent"=0 = a front and a back launch per iteration and lane.  "vectors"=0: the walker ships
    every expanded node's full-precision vector (the reference's data flow); 1: the re-rank reads a packed copy in HBM."""
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, L, with_stats=True)
    ids, dists, st = _run_engine(ix, q, 10, L, graph=0, persistent=persistent, vectors=vectors, pull=0)
    assert st["graph_pull"] == 0
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
    assert st["dist_evals"] == int(st_o[:, 2].sum())
    assert st["fetched"] == int(st_o[:, 3].sum())


@pytest.mark.parametrize("fixture", ["small_f32", "small_u8", "small_deep", "small_i8"])
@pytest.mark.parametrize("persistent", [0, 1])
@pytest.mark.parametrize("L", [10, 64])
def test_engine_device_loop_variants_match_oracle(request, libbang, fixture, persistent, L):
    """Graph resident in HBM, both loop forms: ONE self-paced persistent search kernel per batch ("persistent"=1, the default)
    or a front and a back launch per iteration with the termination flag polled every 16 iterations ("persistent"=0)."""
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, L, with_stats=True)
    ids, dists, st = _run_engine(ix, q, 10, L, graph=1, persistent=persistent, timing=1)
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
    assert st["dist_evals"] == int(st_o[:, 2].sum())
    assert st["candidates"] == int(st_o[:, 1].sum())
    assert st["persistent"] in (0, persistent)          # PQ layouts that need the 256-VGPR build keep the launch-per-iteration loop


@pytest.mark.parametrize("fixture,L", [("small_u8", 40), ("small_u8", 256), ("small_deep", 40), ("small_deep", 200)])
@pytest.mark.parametrize("graph", [0, 1])
def test_engine_long_worklists_and_exact_size_pivot_table(request, libbang, fixture, L, graph):
    """The persistent kernel keeps the pivot table and every wave's merge scratch in LDS: at large L (or for the 152 KB padded
    table of the 96-dims-in-74-chunks layout at any L) the engine switches to the exact-size table; results do not change."""
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    ids_o, dists_o = O.Oracle(ix).search(q, 10, L)
    ids, dists, st = _run_engine(ix, q, 10, L, graph=graph)
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
    ids2, dists2, _ = _run_engine(ix, q, 10, L, graph=graph, pq_ragged=0, persistent=0)
    assert np.array_equal(ids, ids2) and np.array_equal(dists.view(np.uint32), dists2.view(np.uint32))


def test_query_smaller_than_allocation(request, libbang, small_f32):
    """bang_alloc(Q) then bang_query on fewer queries (the vector log and the rank-major distance matrix re-stride)."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_f32
    ids_o, dists_o = O.Oracle(ix).search(q[:17], 10, 30)
    for graph in (0, 1):
        with bang_amd.Engine(ix.dtype, graph=graph, lanes=2) as e:
            e.load_index(ix)
            e.set_searchparams(10, 30)
            e.alloc(q.shape[0])
            e.init(17)
            ids, dists = e.query(q[:17])
            assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))


@pytest.mark.parametrize("fixture", ["small_f32", "small_u8", "small_deep", "small_i8"])
@pytest.mark.parametrize("L", [10, 37, 64, 152, 300, 512])
def test_search_kernel_matches_oracle_per_query(request, libbang, fixture, L):
    """Graph in HBM, the query-resident search kernel ("search"=1): a wave owns one query from its first to its last
    iteration, worklist and survivors stay in LDS, the filter is updated with plain stores.  Ids, distances and the PER-QUERY
    counters (iterations, expanded nodes, distance evaluations, ids offered to the filter) equal the oracle's; the round-1
    loops ("search"=0) return the same bits."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, L, with_stats=True)
    with bang_amd.Engine(ix.dtype, graph=1, search=1, timing=1) as e:
        e.load_index(ix)
        e.set_searchparams(10, L)
        e.alloc(q.shape[0])
        for _ in range(2):
            e.init(q.shape[0])
            ids, dists = e.query(q)
            st = e.stats()
            assert st["search_kernel"] == 1 and st["persistent"] == 1 and st["front_launches"] == 1
            assert np.array_equal(ids, ids_o)
            assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
            assert np.array_equal(e.query_counters(q.shape[0]), st_o)
            assert st["iterations"] == int(st_o[:, 0].max()) and 0 < st["front_ms"]
        e.free()
        e.unload()
    ids0, dists0, st0 = _run_engine(ix, q, 10, L, graph=1, search=0)
    assert st0["search_kernel"] == 0
    assert np.array_equal(ids0, ids_o) and np.array_equal(dists0.view(np.uint32), dists_o.view(np.uint32))


@pytest.mark.parametrize("Q", [1, 2, 63, 700])
def test_search_kernel_batch_sizes_and_handout(request, libbang, small_u8, Q):
    """Fewer queries than waves (spread over the CUs), and more queries than resident waves on a small grid (max 2 workgroups
    via BANG_SEARCH_MAX_WGS: every wave pulls follow-up queries from the hand-out counter)."""
    import os
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    qq = np.ascontiguousarray(np.tile(q, ((Q + q.shape[0] - 1) // q.shape[0], 1))[:Q])
    ids_o, dists_o, st_o = O.Oracle(ix).search(qq, 10, 48, with_stats=True)
    for max_wgs in ("0", "2"):
        os.environ["BANG_SEARCH_MAX_WGS"] = max_wgs
        try:
            with bang_amd.Engine(ix.dtype, graph=1, search=1) as e:
                e.load_index(ix)
                e.set_searchparams(10, 48)
                e.alloc(Q)
                e.init(Q)
                ids, dists = e.query(qq)
                assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
                assert np.array_equal(e.query_counters(Q), st_o)
        finally:
            os.environ.pop("BANG_SEARCH_MAX_WGS", None)


@pytest.mark.parametrize("fixture", ["small_f32", "small_u8", "small_deep", "small_i8"])
@pytest.mark.parametrize("L", [10, 64, 152, 400])
@pytest.mark.parametrize("vectors", [0, 1])
def test_search_kernel_host_paced_matches_oracle_per_query(request, libbang, fixture, L, vectors):
    """Graph in host RAM (the north-star path), host-paced form of the query-resident search kernel: the waves of a workgroup
    publish their parents in one store per round, the C++ walker writes the adjacency rows (and, "vectors"=0, the
    full-precision vectors) through the PCIe BAR.  Per-query counters, ids and distances equal the oracle's."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, L, with_stats=True)
    with bang_amd.Engine(ix.dtype, graph=0, search=1, vectors=vectors, timing=1, pull=0) as e:
        e.load_index(ix)
        e.set_searchparams(10, L)
        e.alloc(q.shape[0])
        for _ in range(2):
            e.init(q.shape[0])
            ids, dists = e.query(q)
            st = e.stats()
            assert st["graph_pull"] == 0
            if not st["search_kernel"]:
                pytest.skip("device memory is not CPU-writable here (no large BAR): the host-paced search kernel is not used")
            assert st["persistent"] == 1 and st["front_launches"] == 1 and st["vectors_on_device"] == vectors
            assert np.array_equal(ids, ids_o)
            assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
            assert np.array_equal(e.query_counters(q.shape[0]), st_o)
            assert st["h2d_bytes"] > 0
        e.free()
        e.unload()


@pytest.mark.parametrize("Q,threads", [(1, 1), (5, 3), (63, 2), (700, 0)])
def test_search_kernel_host_paced_batch_sizes(request, libbang, small_u8, Q, threads):
    import os
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    qq = np.ascontiguousarray(np.tile(q, ((Q + q.shape[0] - 1) // q.shape[0], 1))[:Q])
    ids_o, dists_o, st_o = O.Oracle(ix).search(qq, 10, 48, with_stats=True)
    for max_wgs in ("0", "2"):
        os.environ["BANG_SEARCH_MAX_WGS"] = max_wgs
        try:
            with bang_amd.Engine(ix.dtype, graph=0, search=1, threads=threads, pull=0) as e:
                e.load_index(ix)
                e.set_searchparams(10, 48)
                e.alloc(Q)
                e.init(Q)
                ids, dists = e.query(qq)
                if not e.stats()["search_kernel"]:
                    pytest.skip("no large BAR")
                assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
                assert np.array_equal(e.query_counters(Q), st_o)
        finally:
            os.environ.pop("BANG_SEARCH_MAX_WGS", None)


# ------------------------------------------------------------------------------------------------------------------ pull mode
@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["small_f32", "small_u8", "small_deep", "small_i8"])
@pytest.mark.parametrize("L", [10, 64, 152, 400])
def test_search_kernel_pull_mode_matches_oracle_per_query(request, libbang, fixture, L):
    """Graph in host RAM, PULL mode: the adjacency lists live as 256-byte rows in pinned host memory and the self-paced search
    kernel fetches a parent's row over PCIe by itself -- no walker thread, no bytes pushed by the host.  Per-query counters, ids
    and distances equal the oracle's; one row is pulled per expansion."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    Q = q.shape[0]
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, L, with_stats=True)
    with bang_amd.Engine(ix.dtype, graph=0, pull=1, timing=1) as e:
        e.load_index(ix)
        e.set_searchparams(10, L)
        e.alloc(Q)
        for _ in range(2):
            e.init(Q)
            ids, dists = e.query(q)
            st = e.stats()
            assert st["graph_mode"] == 0 and st["graph_pull"] == 1 and st["search_kernel"] == 1 and st["front_launches"] == 1
            assert st["vectors_on_device"] == 1 and st["h2d_bytes"] == 0 and st["pacing_groups"] == 0
            assert np.array_equal(ids, ids_o)
            assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
            assert np.array_equal(e.query_counters(Q), st_o)
            assert st["pulled_bytes"] == 256 * (int(st_o[:, 1].sum()) - Q)
        e.free()
        e.unload()


@pytest.mark.gpu
@pytest.mark.parametrize("Q", [1, 5, 63, 700])
def test_search_kernel_pull_mode_batch_sizes(libbang, small_u8, Q):
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    qq = np.ascontiguousarray(np.tile(q, ((Q + q.shape[0] - 1) // q.shape[0], 1))[:Q])
    ids_o, dists_o, st_o = O.Oracle(ix).search(qq, 10, 48, with_stats=True)
    with bang_amd.Engine(ix.dtype, graph=0, pull=1) as e:
        e.load_index(ix)
        e.set_searchparams(10, 48)
        e.alloc(Q)
        e.init(Q)
        ids, dists = e.query(qq)
        assert e.stats()["graph_pull"] == 1
        assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
        assert np.array_equal(e.query_counters(Q), st_o)


@pytest.mark.gpu
def test_pull_mode_is_the_default_and_needs_resident_vectors(libbang, small_u8):
    """"auto" takes the pull mode whenever the vectors are resident in HBM and the rows fit the host memory; asked for explicitly
    without resident vectors it is an error, not a silent change of mode."""
    import bang_amd
    ix, q, _, _ = small_u8
    ids0, _, st0 = _run_engine(ix, q, 10, 40, graph=0)
    assert st0["graph_pull"] == 1 and st0["vectors_on_device"] == 1
    ids1, _, st1 = _run_engine(ix, q, 10, 40, graph=0, vectors=0)            # vectors shipped by the walker: no pull
    assert st1["graph_pull"] == 0 and np.array_equal(ids0, ids1)
    with bang_amd.Engine(ix.dtype, graph=0, pull=1, vectors=0) as e:
        with pytest.raises(bang_amd.BangError, match="pull"):
            e.load_index(ix)
    with bang_amd.Engine(ix.dtype, graph=0) as e:
        e.load_index(ix)
        with pytest.raises(bang_amd.BangError, match="before bang_load"):
            e.set_option("pull", 0)


@pytest.mark.gpu
def test_pull_rows_file_is_shared_and_validated(libbang, small_u8, small_i8, tmp_path, monkeypatch):
    """BANG_PULL_ROWS_DIR: the rows live in ONE file per node -- built by the first engine that loads the index, mapped by the
    next ones (here: two engines of one process); a file that belongs to another index is rebuilt, not trusted."""
    import os
    import bang_amd
    from oracle import oracle as O
    monkeypatch.setenv("BANG_PULL_ROWS_DIR", str(tmp_path))
    ix, q, _, _ = small_u8
    ids_o, dists_o = O.Oracle(ix).search(q, 10, 40)
    path = tmp_path / "index_pull_rows.bin"

    def run(index, queries):
        with bang_amd.Engine(index.dtype, graph=0, pull=1) as e:
            e.load_index(index)
            e.set_searchparams(10, 40)
            e.alloc(queries.shape[0])
            e.init(queries.shape[0])
            out = e.query(queries)
            assert e.stats()["graph_pull"] == 1
            return out
    ids, dists = run(ix, q)
    assert path.exists() and path.stat().st_size == ix.N * 256 + 4096
    rows = np.fromfile(path, np.uint32, ix.N * 64).reshape(ix.N, 64)
    deg, adj = ix.degrees(), ix.adjacency()
    for i in (0, ix.N // 2, ix.N - 1):
        assert np.array_equal(rows[i, :deg[i]], adj[i][:deg[i]]) and (rows[i, deg[i]:] == 0xFFFFFFFF).all()
    assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
    ino = path.stat().st_ino
    ids2, _ = run(ix, q)                                                       # second engine: maps the existing file
    assert path.stat().st_ino == ino and np.array_equal(ids2, ids_o)
    # same name and size, other content: must be rebuilt
    blob = bytearray(path.read_bytes())
    blob[: ix.N * 256] = b"\x00" * (ix.N * 256)
    blob[ix.N * 256 + 2048 + 33] ^= 0xFF                                       # break the sample hash of the signature
    path.write_bytes(bytes(blob))
    ids3, _ = run(ix, q)
    assert np.array_equal(ids3, ids_o) and np.array_equal(np.fromfile(path, np.uint32, ix.N * 64).reshape(ix.N, 64), rows)
    assert not [f for f in os.listdir(tmp_path) if ".tmp." in f]
    # the signature covers EVERY adjacency list: an index edited in place (same N, R, medoid) gets its own rows
    import copy
    ix2 = copy.copy(ix)
    ix2.graph = np.array(ix.graph, copy=True)
    vb = ix.D * (4 if ix.dtype == "float" else 1)
    node = 2049
    adj2 = ix2.graph[node, vb + 4: vb + 4 + 4 * ix.R].view(np.uint32)
    deg2 = int(ix2.graph[node, vb: vb + 4].view(np.uint32)[0])
    repl = next(c for c in range(ix.N) if c != node and c not in set(adj2[:deg2].tolist()))
    adj2[0] = repl
    adj2[:deg2] = np.sort(adj2[:deg2])
    ids4, _ = run(ix2, q)
    assert np.array_equal(ids4, O.Oracle(ix2).search(q, 10, 40)[0])
    rows = np.fromfile(path, np.uint32, ix.N * 64).reshape(ix.N, 64)
    assert repl in rows[node, :deg2].tolist()


@pytest.mark.gpu
def test_pull_form_failures_are_reported_not_faults(libbang, small_u8, tmp_path, monkeypatch):
    """The self-paced pull form has no host thread in its loop, so what can go wrong with the rows it reads by itself is caught at
    both ends: (1) rows overwritten behind the engine's back (ids out of range) end the batch with an error naming the cause -- no
    wild read of the code table, no hang -- and the engine answers correctly again once the rows are back; (2) a rows file
    truncated between bang_load and bang_alloc is refused at bang_alloc."""
    import os
    import bang_amd
    from oracle import oracle as O
    monkeypatch.setenv("BANG_PULL_ROWS_DIR", str(tmp_path))
    ix, q, _, _ = small_u8
    ids_o, dists_o = O.Oracle(ix).search(q, 10, 40)
    path = tmp_path / "index_pull_rows.bin"
    with bang_amd.Engine(ix.dtype, graph=0, pull=1) as e:
        e.load_index(ix)
        e.set_searchparams(10, 40)
        e.alloc(q.shape[0])
        e.init(q.shape[0])
        ids, _ = e.query(q)
        assert np.array_equal(ids, ids_o) and e.stats()["graph_pull"] == 1
        # (1) another process scribbles over the rows: the mapping is shared, the kernel reads what is there now
        rows = np.memmap(path, np.uint32, "r+", shape=(ix.N, 64))
        saved = np.array(rows[:, 0])
        rows[:, 0] = np.uint32(ix.N + 7)
        rows.flush()
        e.init(q.shape[0])
        with pytest.raises(bang_amd.BangError, match="out of range"):
            e.query(q)
        rows[:, 0] = saved
        rows.flush()
        e.init(q.shape[0])
        ids2, dists2 = e.query(q)
        assert np.array_equal(ids2, ids_o) and np.array_equal(dists2.view(np.uint32), dists_o.view(np.uint32))
        del rows
        # (2) the file shrinks between bang_load and the next bang_alloc
        e.free()
        size = path.stat().st_size
        os.truncate(path, size - 4096 - 256 * 10)
        with pytest.raises(bang_amd.BangError, match="truncated or replaced"):
            e.alloc(q.shape[0])
        os.truncate(path, size)                      # (grown back: the pages behind the old end read as zeros, the signature is gone)
        with pytest.raises(bang_amd.BangError, match="signature"):
            e.alloc(q.shape[0])
        e.unload()


# -------------------------------------------------------------------------------------------------------------- streamed load
def _entry_source(ix):
    """A Python entry source over an in-memory index: copies the requested node range in the reference entry layout."""
    import ctypes as C
    graph = np.ascontiguousarray(ix.graph, dtype=np.uint8)

    def src(first, count, dst):
        C.memmove(dst, graph[first:first + count].ctypes.data, count * ix.entry_len)
        return 0
    return src


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["small_f32", "small_u8", "small_i8"])
def test_streamed_load_matches_oracle(request, libbang, fixture):
    """bang_load_stream_e: the graph entries pass through in chunks (vectors -> HBM, adjacency lists -> pull rows) and are never
    resident as a whole; the search runs in pull mode and returns the oracle's bits."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    Q = q.shape[0]
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, 48, with_stats=True)
    for graph in (bang_amd.GRAPH_HOST, bang_amd.GRAPH_AUTO):
        with bang_amd.Engine(ix.dtype, graph=graph) as e:
            e.load_stream(ix, _entry_source(ix))
            e.set_searchparams(10, 48)
            e.alloc(Q)
            e.init(Q)
            ids, dists = e.query(q)
            st = e.stats()
            assert st["graph_mode"] == 0 and st["graph_pull"] == 1 and st["vectors_on_device"] == 1
            assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
            assert np.array_equal(e.query_counters(Q), st_o)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["small_u8", "small_deep", "small_i8"])
@pytest.mark.parametrize("threads", [0, 1, 3])
def test_walker_serves_from_the_pull_rows(request, libbang, fixture, threads):
    """Option walker = 1: the north-star data flow (C++ walker threads hand the parents' adjacency lists to the host-paced search kernel,
    bang_search.cu:771-813) on an index loaded in pull mode -- the threads read the 256-byte pull rows, not graph entries, so the form also
    runs on a STREAMED load, which keeps no graph image at all.  Same bits as the oracle, per query."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    Q = q.shape[0]
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, 48, with_stats=True)
    for load in ("memory", "stream"):
        with bang_amd.Engine(ix.dtype, graph=bang_amd.GRAPH_HOST, walker=1, threads=threads) as e:
            if load == "stream":
                e.load_stream(ix, _entry_source(ix))
            else:
                e.load_index(ix)
            e.set_searchparams(10, 48)
            e.alloc(Q)
            for _ in range(2):
                e.init(Q)
                ids, dists = e.query(q)
                st = e.stats()
                assert st["walker_rows"] == 1 and st["search_kernel"] == 1 and st["graph_pull"] == 0 and st["pacing_groups"] > 0, (load, st)
                assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), load
                assert np.array_equal(e.query_counters(Q), st_o), load
            e.free(); e.unload()


@pytest.mark.gpu
def test_walker_form_serves_the_rows_it_holds_in_hbm_itself(libbang, small_u8, monkeypatch):
    """Walker-from-rows with part of the rows also in HBM (bang_rows_slice_e: rows [0, N/2)): a wave whose parent's row is there loads it
    itself and tells the walker threads there is nothing to fetch for it; the other rows still come from the threads.  Same bits either
    way (BANG_WALKER_SELF_ROWS=0 sends every row through the threads), and the stats say how many rows stayed on the GPU."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    Q = q.shape[0]
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, 48, with_stats=True)
    own = {}
    for self_rows in ("1", "0"):
        monkeypatch.setenv("BANG_WALKER_SELF_ROWS", self_rows)
        with bang_amd.Engine(ix.dtype, graph=bang_amd.GRAPH_HOST, walker=1, threads=4) as e:
            e.load_index(ix)
            e.rows_slice(0, ix.N // 2)
            e.set_searchparams(10, 48)
            e.alloc(Q)
            e.init(Q)
            ids, dists = e.query(q)
            st = e.stats()
            assert st["walker_rows"] == 1 and st["graph_pull"] == 0, st
            assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), self_rows
            assert np.array_equal(e.query_counters(Q), st_o), self_rows
            own[self_rows] = st["rows_from_own_hbm"]
            e.free(); e.unload()
    assert own["0"] == 0 and 0 < own["1"] < int(st["candidates"]), own


@pytest.mark.gpu
def test_peer_slice_shorter_than_its_slot_is_refused_and_moved_slice_stats(libbang, small_u8):
    """ADVICE r5.  (1) bang_rows_import_e takes the row count the exporter reported: a peer allocation shorter than the slot it is to serve is
    refused before it is mapped (the kernel reads base + p * 256 for every p of the slot).  (2) A slice moved off row 0 without a slice table is
    not reachable by the kernel: the stats say 0 rows in HBM / from own HBM, and every row counted as pulled."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    Q = q.shape[0]
    n = ix.N // 2
    with bang_amd.Engine(ix.dtype, graph=bang_amd.GRAPH_HOST) as e:
        e.load_index(ix)
        e.rows_slice(0, n)
        h, first, rows = e.rows_export()
        assert (first, rows) == (0, n) and any(h)
        e.rows_import(0, 2, n, None)
        with pytest.raises(bang_amd.BangError, match="exported with"):
            e.rows_import(1, 2, n, h, rows=n - 1)                     # (slot 1 spans rows [n, 2n): N - n >= n of them)
        with pytest.raises(bang_amd.BangError, match="exported with"):
            e.rows_import(1, 2, n, h)                                 # no count at all
        e.unload()
    ids_o, _ = O.Oracle(ix).search(q, 10, 48)
    with bang_amd.Engine(ix.dtype, graph=bang_amd.GRAPH_HOST) as e:
        e.load_index(ix)
        e.rows_slice(n, n)                                            # rows [n, 2n) in HBM, no table: the kernel pulls everything
        e.set_searchparams(10, 48)
        e.alloc(Q)
        e.init(Q)
        ids, _ = e.query(q)
        st = e.stats()
        assert np.array_equal(ids, ids_o)
        assert st["graph_pull"] == 1 and st["rows_in_hbm"] == 0 and st["rows_from_own_hbm"] == 0 and st["rows_from_peer"] == 0, st
        assert st["pulled_bytes"] == 256 * (int(st["candidates"]) - Q), st
        e.free(); e.unload()


@pytest.mark.gpu
def test_streamed_load_needs_the_pull_mode(libbang, small_u8):
    """Without a resident graph nothing but the pull mode can run: a streamed load refuses configurations that exclude it, and a
    walker form asked for afterwards is an error (there is no file to map)."""
    import bang_amd
    ix, q, _, _ = small_u8
    for opts in (dict(graph=1), dict(graph=0, pull=0), dict(graph=0, vectors=0), dict(graph=0, persistent=0), dict(graph=0, search=0)):
        with bang_amd.Engine(ix.dtype, **opts) as e:
            with pytest.raises(bang_amd.BangError, match="pull mode"):
                e.load_stream(ix, _entry_source(ix))
    with bang_amd.Engine(ix.dtype, graph=0) as e:
        e.load_stream(ix, _entry_source(ix))
        e.set_option("persistent", 0)
        e.set_searchparams(10, 40)
        with pytest.raises(bang_amd.BangError, match="streamed"):
            e.alloc(q.shape[0])
    with bang_amd.Engine(ix.dtype, graph=0) as e:                   # a failing source fails the load
        with pytest.raises(bang_amd.BangError, match="entry source"):
            e.load_stream(ix, lambda first, count, dst: 1)


@pytest.mark.gpu
def test_file_load_streams_the_graph_and_maps_it_on_demand(libbang, small_u8, tmp_path, monkeypatch):
    """bang_load on the host placement: `_disk.bin` is streamed (never resident) when the pull mode applies; a walker form chosen
    after the load maps the file then; BANG_STREAM_LOAD=0 maps it up front as before.  Same bits every way."""
    import bang_amd
    from bang_amd import formats
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    Q = q.shape[0]
    prefix = str(tmp_path / "idx")
    formats.write_index(prefix, ix)
    ids_o, dists_o = O.Oracle(ix).search(q, 10, 40)

    def run(late_opts=None, **opts):
        with bang_amd.Engine(ix.dtype, graph=0, **opts) as e:
            e.load(prefix)
            for k, v in (late_opts or {}).items():
                e.set_option(k, v)
            e.set_searchparams(10, 40)
            e.alloc(Q)
            e.init(Q)
            ids, dists = e.query(q)
            st = e.stats()
        assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
        return st
    assert run()["graph_pull"] == 1                                   # streamed, pulled
    st = run(late_opts=dict(persistent=0))                            # streamed at load; the walker loop maps the file at bang_alloc
    assert st["graph_pull"] == 0 and st["persistent"] == 0 and st["h2d_bytes"] > 0
    assert run(pull=0)["graph_pull"] == 0                             # walker asked for up front: mapped at load
    monkeypatch.setenv("BANG_STREAM_LOAD", "0")
    assert run()["graph_pull"] == 1                                   # mapped at load, rows built from the mapping


@pytest.mark.gpu
def test_streamed_load_shares_and_checks_the_rows_file(libbang, small_u8, small_i8, tmp_path, monkeypatch):
    import bang_amd
    from oracle import oracle as O
    monkeypatch.setenv("BANG_PULL_ROWS_DIR", str(tmp_path))
    ix, q, _, _ = small_u8
    ids_o, _ = O.Oracle(ix).search(q, 10, 40)
    path = tmp_path / "index_pull_rows.bin"

    def run(index):
        with bang_amd.Engine(index.dtype, graph=0) as e:
            e.load_stream(index, _entry_source(index))
            e.set_searchparams(10, 40)
            e.alloc(q.shape[0])
            e.init(q.shape[0])
            return e.query(q)[0]
    assert np.array_equal(run(ix), ids_o) and path.exists()
    ino = path.stat().st_ino
    assert np.array_equal(run(ix), ids_o) and path.stat().st_ino == ino       # second load: rows mapped, signature checked at the end
    good = path.read_bytes()
    blob = bytearray(good)
    blob[ix.N * 256 + 2048 + 33] ^= 0xFF
    blob[:256] = b"\x00" * 256                                                 # (and a wrong first row)
    path.write_bytes(bytes(blob))
    assert np.array_equal(run(ix), ids_o) and path.read_bytes() == good       # stale file: removed, the entries stream through once more
    # an index EDITED IN PLACE (same N, R, medoid; one adjacency list differs, far from any sampled node) must not inherit the rows
    import copy
    ix2 = copy.copy(ix)
    ix2.graph = np.array(ix.graph, copy=True)
    vb = ix.D * (4 if ix.dtype == "float" else 1)
    node = 1237
    adj = ix2.graph[node, vb + 4: vb + 4 + 4 * ix.R].view(np.uint32)
    deg = int(ix2.graph[node, vb: vb + 4].view(np.uint32)[0])
    repl = next(c for c in range(ix.N) if c != node and c not in set(adj[:deg].tolist()))
    adj[0] = repl
    adj[:deg] = np.sort(adj[:deg])
    ids_o2, _ = O.Oracle(ix2).search(q, 10, 40)
    assert np.array_equal(run(ix2), ids_o2) and path.read_bytes() != good
    rows = np.fromfile(path, np.uint32, ix.N * 64).reshape(ix.N, 64)
    assert repl in rows[node, :deg].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["small_u8", "small_f32", "small_i8"])
def test_bang_load_reads_a_diskann_index_directly(request, libbang, fixture, tmp_path):
    """No `_disk.bin` / `_disk_metadata.bin`: bang_load takes DiskANN's own sector-padded `<p>_disk.index` (shuffled adjacency
    lists, garbage behind them) and does the reference's preprocessing on the fly -- streamed in pull mode, converted into a
    resident copy for the placements that walk or upload the whole graph.  Same bits as the oracle on the converted index."""
    import os
    import bang_amd
    from bang_amd import formats
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    Q = q.shape[0]
    prefix = str(tmp_path / "raw")
    formats.write_index(prefix, ix)
    os.remove(prefix + "_disk.bin")
    os.remove(prefix + "_disk_metadata.bin")
    formats.write_diskann_index(prefix + "_disk.index", ix.vectors(), ix.degrees(), ix.adjacency(), ix.medoid, pad_garbage=True)
    ids_o, dists_o = O.Oracle(ix).search(q, 10, 40)
    for opts, pulled in ((dict(graph=0), 1), (dict(graph=0, pull=0), 0), (dict(graph=1), 0), (dict(graph=0, persistent=0), 0)):
        with bang_amd.Engine(ix.dtype, **opts) as e:
            e.load(prefix)
            e.set_searchparams(10, 40)
            e.alloc(Q)
            e.init(Q)
            ids, dists = e.query(q)
            assert e.stats()["graph_pull"] == pulled, opts
        assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), opts
    wrong = "float" if ix.dtype != "float" else "uint8"
    with bang_amd.Engine(wrong, graph=0) as e:                      # the record length does not fit the element size: refused
        with pytest.raises(bang_amd.BangError):
            e.load(prefix)


def test_query_dev_leaves_identical_results_in_device_buffers(libbang, small_u8):
    """bang_query_dev_e: ids [Q][k] / dists [k][Q] written to the caller's DEVICE buffers equal what bang_query_e returns to the host
    (both placements; the N > 1 job all-gathers straight from such a buffer)."""
    import bang_amd
    from bang_amd import binding as B
    ix, q, _, _ = small_u8
    Q, k, L = q.shape[0], 10, 48
    for graph in (bang_amd.GRAPH_HOST, bang_amd.GRAPH_DEVICE):
        with bang_amd.Engine("uint8", graph=graph) as e:
            e.load_index(ix)
            e.set_searchparams(k, L)
            e.alloc(Q)
            e.init(Q)
            ids, dists = e.query(q)
            d_ids, d_dists = B.DeviceBuffer(Q * k * 8), B.DeviceBuffer(Q * k * 4)
            for with_dists in (True, False):
                e.init(Q)
                e.query_dev(q, d_ids.ptr, d_dists.ptr if with_dists else 0)
                assert np.array_equal(d_ids.download(np.uint64, (Q, k)), ids)
                if with_dists:
                    assert np.array_equal(d_dists.download(np.float32, (k, Q)).view(np.uint32), dists.view(np.uint32))
            # a shorter batch on the same allocation
            e.init(Q // 2)
            e.query_dev(q[: Q // 2], d_ids.ptr, d_dists.ptr)
            assert np.array_equal(d_ids.download(np.uint64, (Q, k))[: Q // 2], ids[: Q // 2])
            d_ids.free(); d_dists.free()
            e.free()
            e.unload()


def test_host_paced_hand_shake_failure_is_reported_and_survivable(libbang, small_u8):
    """The abort path of the host-paced search kernel (the reference's loop, bang_search.cu:771-838, would block forever).  The
    walker team is stalled by the test hook for longer than kernel_go_timeout_ms: every pacing group gives up on its own, sets the
    abort word and leaves; the walker, back from its stall, finds nobody to serve and gives up after host_walk_timeout_ms;
    bang_query reports BANG_ERR_HIP naming the side that gave up first, the process lives, and the NEXT bang_init + bang_query on
    the same allocation returns the oracle's answer.  Both timeouts are run-time options (range-checked)."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    ids_o, dists_o = O.Oracle(ix).search(q, 10, 40)
    with bang_amd.Engine(ix.dtype, graph=0, search=1, pull=0, threads=3) as e:
        e.load_index(ix)
        e.set_searchparams(10, 40)
        e.alloc(q.shape[0])
        e.init(q.shape[0])
        ids, dists = e.query(q)
        assert e.stats()["search_kernel"] == 1 and e.stats()["graph_pull"] == 0
        assert np.array_equal(ids, ids_o)
        e.set_option("kernel_go_timeout_ms", 150)
        e.set_option("host_walk_timeout_ms", 400)
        e.set_option("walker_stall_ms", 900)
        e.init(q.shape[0])
        with pytest.raises(bang_amd.BangError, match="gave up waiting for the host walker"):
            e.query(q)
        # the allocation is still good: state is rebuilt by bang_init, the hand-shake words by the next query
        e.set_option("kernel_go_timeout_ms", 30000)
        e.set_option("host_walk_timeout_ms", 20000)
        for _ in range(2):
            e.init(q.shape[0])
            ids, dists = e.query(q)
            assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
        with pytest.raises(bang_amd.BangError):
            e.set_option("kernel_go_timeout_ms", 1)              # out of range
        e.free()
        e.unload()


@pytest.mark.parametrize("fixture", ["small_u8", "small_deep", "small_f32"])
def test_code_row_stride_does_not_change_results(request, libbang, fixture):
    """Option code_stride: the PQ code rows in HBM packed as in the file (0), padded by the engine (auto: 70 / 74 -> 128; 32 stays), or
    at an explicit stride (dword aligned or not) -- and a caller's own padded device table -- all return the oracle's answer in every
    loop form."""
    import bang_amd
    from bang_amd import binding as B
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    Q, k, L = q.shape[0], 10, 40
    ids_o, dists_o = O.Oracle(ix).search(q, k, L)
    want_auto = 128 if ix.m > 64 else ix.m

    def run(**opts):
        d_codes = opts.pop("d_codes", None)
        ext_stride = opts.pop("ext_stride", 0)
        with bang_amd.Engine(ix.dtype, **opts) as e:
            e.load_index(ix, d_codes=d_codes, code_stride=ext_stride)
            e.set_searchparams(k, L)
            e.alloc(Q)
            e.init(Q)
            ids, dists = e.query(q)
            st = e.stats()
            e.free()
            e.unload()
        assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), opts
        return st
    for graph in (0, 1):
        assert run(graph=graph)["code_stride"] == want_auto
        assert run(graph=graph, code_stride=0)["code_stride"] == ix.m
        assert run(graph=graph, code_stride=ix.m + 3)["code_stride"] == ix.m + 3
    assert run(graph=0, pull=0, code_stride=128)["code_stride"] == 128                # host-paced search kernel
    assert run(graph=0, persistent=0, code_stride=128)["code_stride"] == 128          # a launch per iteration
    assert run(graph=1, search=0, code_stride=128)["code_stride"] == 128
    padded = np.full((ix.N, 256), 0x5A, np.uint8)
    padded[:, : ix.m] = ix.codes
    buf = B.DeviceBuffer.from_numpy(padded, slack=256)
    assert run(graph=1, d_codes=buf.ptr, ext_stride=256)["code_stride"] == 256         # the caller's layout is taken as it is
    buf.free()
    with pytest.raises(bang_amd.BangError):
        run(graph=1, code_stride=ix.m - 1)


def test_shared_load_from_a_sibling_engines_vectors_and_rows_file(libbang, small_u8, tmp_path, monkeypatch):
    """bang_load_shared_e: a second engine (standing in for another rank of the node) loads WITHOUT reading an index entry -- the
    vectors are copied device to device out of the buffer the first engine's streamed load filled (desc->d_vectors), the adjacency
    rows come from the node's rows file, checked against the hash the first engine reports.  Same answers as the oracle; a wrong
    hash is refused and leaves the rows file alone."""
    import bang_amd
    from bang_amd import binding as B
    from oracle import oracle as O
    monkeypatch.setenv("BANG_PULL_ROWS_DIR", str(tmp_path))
    ix, q, _, _ = small_u8
    Q, k, L = q.shape[0], 10, 40
    ids_o, dists_o = O.Oracle(ix).search(q, k, L)
    vb = ix.D
    v1 = B.DeviceBuffer(ix.N * vb + 256)
    v2 = B.DeviceBuffer(ix.N * vb + 256)

    def search(e):
        e.set_searchparams(k, L)
        e.alloc(Q)
        e.init(Q)
        out = e.query(q)
        assert e.stats()["graph_pull"] == 1
        return out
    with bang_amd.Engine("uint8", graph=0) as e1:
        e1.load_stream(ix, _entry_source(ix), d_vectors=v1.ptr)
        h = e1.rows_hash()
        ids, dists = search(e1)
        assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
        v2.upload(v1.download(np.uint8, (ix.N * vb,)))                  # (the xGMI broadcast of the multi-GPU job)
        with bang_amd.Engine("uint8", graph=0) as e2:
            with pytest.raises(bang_amd.BangError):
                e2.load_shared(ix, v2.ptr, h ^ 1)
            assert (tmp_path / "index_pull_rows.bin").exists()
            e2.load_shared(ix, v2.ptr, h)
            ids2, dists2 = search(e2)
            assert np.array_equal(ids2, ids_o) and np.array_equal(dists2.view(np.uint32), dists_o.view(np.uint32))
            with pytest.raises(bang_amd.BangError):                     # nothing but the pull mode can run on such an index
                e2.free(); e2.set_option("persistent", 0); e2.alloc(Q)
    v1.free(); v2.free()


@pytest.mark.parametrize("fixture", ["small_u8", "small_f32"])
def test_rows_partly_in_hbm_do_not_change_results(request, libbang, fixture, monkeypatch):
    """Option rows_hbm: in pull mode the adjacency rows of the first nodes also sit in HBM and are read from there; the rest is
    pulled over PCIe.  Same ids / distances / per-query counters as the oracle; pulled_bytes counts the PCIe rows only."""
    import bang_amd
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    Q, k, L = q.shape[0], 10, 40
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, k, L, with_stats=True)

    def run(**opts):
        with bang_amd.Engine(ix.dtype, graph=0, **opts) as e:
            e.load_index(ix)
            e.set_searchparams(k, L)
            e.alloc(Q)
            e.init(Q)
            ids, dists = e.query(q)
            st = e.stats()
            cnt = e.query_counters(Q)
            e.free()
            e.unload()
        assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
        assert np.array_equal(cnt, st_o)
        return st
    base = run()                                                    # auto: a small index is left on the host as asked
    assert base["graph_pull"] == 1 and base["rows_in_hbm"] == 0 and base["pulled_bytes"] == (base["candidates"] - Q) * 256
    monkeypatch.setenv("BANG_ROWS_HBM_MAX_ROWS", str(ix.N // 3))
    part = run(rows_hbm=64)
    assert part["rows_in_hbm"] == ix.N // 3 and 0 < part["pulled_bytes"] < base["pulled_bytes"]
    monkeypatch.delenv("BANG_ROWS_HBM_MAX_ROWS")
    full = run(rows_hbm=64)
    assert full["rows_in_hbm"] == ix.N and full["pulled_bytes"] == 0


def test_candidate_log_hook(libbang, small_u8):
    """bang_get_candidate_log: the nodes every query expanded, in expansion order ([0] = MEDOID, bang_search.cu:452-464,1451-1458) --
    lengths = the per-query candidate counters, every returned id among them (the re-rank only sees expanded nodes)."""
    import bang_amd
    ix, q, _, _ = small_u8
    with bang_amd.Engine(ix.dtype, graph=1) as e:
        e.load_index(ix)
        e.set_searchparams(10, 48)
        e.alloc(q.shape[0])
        e.init(q.shape[0])
        ids, _ = e.query(q)
        cand, cnt = e.candidate_log(q.shape[0], 48)
        ctr = e.query_counters(q.shape[0])
        with pytest.raises(bang_amd.BangError):            # buffers for fewer queries than the batch had: refused, not overrun
            e.candidate_log(q.shape[0] - 1, 48)
        e.free(); e.unload()
    assert np.array_equal(cnt.astype(np.int64), ctr[:, 1]) and (cand[:, 0] == ix.medoid).all()
    for i in range(q.shape[0]):
        log = set(cand[i, :cnt[i]].tolist())
        assert len(log) == cnt[i] and all(int(x) in log for x in ids[i])


@pytest.mark.parametrize("summ_iters", ["-1", "1", "7"])
@pytest.mark.parametrize("graph", [0, 1])
def test_filter_summary_policy_does_not_change_results(libbang, small_u8, monkeypatch, summ_iters, graph):
    """The on-chip filter summary may serve all of a query's iterations (-1: what a full chip gets), none but the first (1: what
    bang_k_search picks for launches of at most 6 queries per CU) or any prefix: a probe it does not answer is simply loaded."""
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    monkeypatch.setenv("BANG_SUMM_ITERS", summ_iters)
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, 64, with_stats=True)
    ids, dists, st = _run_engine(ix, q, 10, 64, graph=graph, search=1)
    assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
    assert st["dist_evals"] == int(st_o[:, 2].sum()) and st["fetched"] == int(st_o[:, 3].sum())
    assert (st["filter_loads_skipped"] > 0) == (summ_iters != "1") or summ_iters == "1"
