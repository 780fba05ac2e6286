"""Kernel-level parity on the GPU: every C-ABI kernel entry (include/bang_c.h section 2) is driven
on its own and compared bit-for-bit with the oracle's restatement of the reference kernel it
replaces.  The oracle plays the host walker (it owns the graph)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BIG = np.float32(3.402823E+38)


class HostQuery:
    """Per-query oracle-side state (bloom as bytes like the reference, worklist, candidate log)."""

    def __init__(self, O, ix):
        self.bloom = np.zeros(O.BF_MEMORY, dtype=np.uint8)
        self.w = (np.zeros(0, np.uint32), np.zeros(0, np.float32), np.zeros(0, np.uint8))
        self.mark = 0x01010101
        self.cand = [ix.medoid]
        self.parent = None
        self.S = np.zeros(0, np.uint32)
        self.d = np.zeros(0, np.float32)


def _adj(ix, node):
    deg = int(ix.degrees()[node])
    return ix.adjacency()[node][:deg]


@pytest.mark.parametrize("fixture,use_lut", [("small_f32", False), ("small_u8", False), ("small_deep", False),
                                             ("small_i8", False), ("small_u8", True), ("small_f32", True)])
def test_stage_by_stage(request, libbang, fixture, use_lut):
    import bang_amd
    from bang_amd.binding import IterState, NO_PARENT, IDLE_PARENT
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    Q, L, iters = 16, 24, 14
    q = q[:Q]
    orc = O.Oracle(ix)
    st = IterState(ix, q, L, use_lut=use_lut)
    luts = [orc.lut_build(q[i]) for i in range(Q)]
    if use_lut:   # K1
        assert np.array_equal(st.lut().view(np.uint32), np.stack(luts).view(np.uint32))
    hq = [HostQuery(O, ix) for _ in range(Q)]
    deg_adj = ix.adjacency(), ix.degrees()
    seed = np.concatenate([[ix.medoid], deg_adj[0][ix.medoid][: int(deg_adj[1][ix.medoid])]]).astype(np.uint32)

    for it in range(1, iters + 1):
        st.iter, st.first = it, 1 if it == 1 else 0
        # ---- K5 filter
        lists = [seed if it == 1 else (_adj(ix, h.parent) if h.parent is not None else np.zeros(0, np.uint32)) for h in hq]
        if it > 1:
            st.stage(lists)
        st.run("filter")
        cnt, ids, _ = st.nbrs()
        for i, h in enumerate(hq):
            h.S = O.filter_ids(h.bloom, lists[i])
            assert cnt[i] == len(h.S), (it, i)
            assert np.array_equal(ids[i, :cnt[i]], h.S), (it, i)
        # ---- K2 distances
        st.run("pqdist")
        _, _, dist = st.nbrs()
        for i, h in enumerate(hq):
            h.d = orc.pqdist(luts[i], h.S)
            assert np.array_equal(dist[i, :len(h.S)].view(np.uint32), h.d.view(np.uint32)), (it, i)
        # ---- K4 parent
        st.run("parent")
        par, mark = st.parents()
        _, _, _, wvis = st.worklist()
        ccnt, cids, crow = st.candidates()
        for i, h in enumerate(hq):
            if it == 1:
                ok, p, mk = O.parent1(h.S, h.d, ix.medoid)
            else:
                ok, p, mk, vis = O.parent2(h.S, h.d, h.w[0], h.w[1], h.w[2], ix.medoid, h.mark)
                h.w = (h.w[0], h.w[1], vis)
            if ok:
                h.parent, h.mark = p, mk
                h.cand.append(p)
                assert par[i] == p, (it, i)
            else:
                h.parent = None
                assert par[i] == (IDLE_PARENT if len(h.S) else NO_PARENT), (it, i)
            assert mark[i] == h.mark, (it, i)
            assert ccnt[i] == len(h.cand) and np.array_equal(cids[i, :ccnt[i]], np.array(h.cand, np.uint32)), (it, i)
            assert np.array_equal(wvis[i, :len(h.w[2])], h.w[2]), (it, i)
        # ---- K3a + K3b sort + merge
        st.run("back")
        wn, wi, wd, wv = st.worklist()
        for i, h in enumerate(hq):
            s_ids, s_d = O.sort_pairs(h.S, h.d)
            h.w = O.merge(s_ids, s_d, it, h.w[0], h.w[1], h.w[2], L, ix.medoid, h.mark)
            n = len(h.w[0])
            assert wn[i] == n, (it, i)
            assert np.array_equal(wi[i, :n], h.w[0]) and np.array_equal(wv[i, :n], h.w[2]), (it, i)
            assert np.array_equal(wd[i, :n].view(np.uint32), h.w[1].view(np.uint32)), (it, i)

    # ---- K6 + K7 re-rank over the logged candidates
    ids_g, dists_g = st.rerank(10)
    for i, h in enumerate(hq):
        e = np.array([orc.exact_dist(c, q[i]) for c in h.cand], dtype=np.float32)
        ids_o, d_o = O.topk(np.array(h.cand, np.uint32), e, 10)
        assert np.array_equal(ids_g[i], ids_o), i
        assert np.array_equal(dists_g[:, i].view(np.uint32), d_o.view(np.uint32)), i


def test_filter_snapshot_and_order(libbang, small_u8):
    """CANON semantics of K5: duplicates inside one batch all pass (snapshot test), survivors keep input
    order, a second offer of the same ids is rejected entirely."""
    from bang_amd.binding import IterState
    ix, q, _, _ = small_u8
    st = IterState(ix, q[:4], 16)
    st.first, st.iter = 0, 2
    lists = [np.array([5, 9, 5, 1234, 9], np.uint32), np.arange(64, dtype=np.uint32)[::-1].copy(),
             np.zeros(0, np.uint32), np.array([4294967295 % ix.N], np.uint32)]
    st.stage(lists)
    st.run("filter")
    cnt, ids, _ = st.nbrs()
    for i, l in enumerate(lists):
        assert cnt[i] == len(l) and np.array_equal(ids[i, :cnt[i]], l)
    st.run("filter")
    cnt, _, _ = st.nbrs()
    assert not cnt.any()


def test_pqdist_float_order_is_canonical(libbang, small_u8):
    """The 8-way strided partial sums + pairwise tree of compute_neighborDist_par differ from a plain
    left-to-right sum in the last bits; the kernel must reproduce the canonical order, not 'a' sum."""
    from bang_amd.binding import IterState
    from oracle import oracle as O
    ix, q, _, _ = small_u8
    orc = O.Oracle(ix)
    st = IterState(ix, q[:8], 16)
    st.first, st.iter = 0, 2
    rng = np.random.default_rng(5)
    lists = [rng.choice(ix.N, 64, replace=False).astype(np.uint32) for _ in range(8)]
    st.stage(lists)
    st.run("filter")
    st.run("pqdist")
    cnt, ids, dist = st.nbrs()
    naive_differs = 0
    for i in range(8):
        lut = orc.lut_build(q[i])
        want = orc.pqdist(lut, ids[i, :cnt[i]])
        assert np.array_equal(dist[i, :cnt[i]].view(np.uint32), want.view(np.uint32))
        for j in range(cnt[i]):
            row = ix.codes[ids[i, j]]
            acc = np.float32(0)
            for c in range(ix.m):
                acc = np.float32(acc + lut[c, row[c]])
            naive_differs += int(acc != want[j])
    assert naive_differs > 0


@pytest.mark.parametrize("fixture,nhi", [("small_u8", 58), ("small_deep", 22)])
def test_exact_size_pivot_table_gives_the_same_distances(request, libbang, fixture, nhi):
    """PQ layouts with 2-dim then 1-dim chunks (128 dims in 70 chunks, 96 in 74): the fused kernel's instance for the
    exact-size ("ragged") pivot table [nhi][256][2] + [m-nhi][256][1] must reproduce the padded table's -- i.e. the
    oracle's -- distances bit for bit."""
    from bang_amd.binding import IterState
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    orc = O.Oracle(ix)
    rng = np.random.default_rng(7)
    lists = [rng.choice(ix.N, 64, replace=False).astype(np.uint32) for _ in range(8)]
    got = {}
    for ragged in (False, True):
        st = IterState(ix, q[:8], 16, ragged=ragged)
        assert st.pq_nhi == (nhi if ragged else 0)
        st.first, st.iter = 0, 2
        st.stage(lists)
        st.run("front")
        got[ragged] = st.nbrs()
    for a, b in zip(got[False], got[True]):
        assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    cnt, ids, dist = got[True]
    for i in range(8):
        want = orc.pqdist(orc.lut_build(q[i]), ids[i, :cnt[i]])
        assert np.array_equal(dist[i, :cnt[i]].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("fixture,L,ragged", [("small_f32", 40, False), ("small_i8", 24, False), ("small_u8", 32, False),
                                              ("small_deep", 32, True)])
def test_search_kernel_alone(request, libbang, fixture, L, ragged):
    """bang_k_search called directly (self-paced form: graph in HBM, no host in the loop), then K6 + K7: the ids and exact
    distances must be the oracle's for the whole search, the candidate log its expansion order, the iteration counts its own."""
    from bang_amd.binding import IterState
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    q = q[:24]
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, 10, L, with_stats=True)
    st = IterState(ix, q, L, device_graph=True, ragged=ragged)
    iters = st.run_search()
    assert np.array_equal(iters.astype(np.int64), st_o[:, 0])
    ccnt, _, _ = st.candidates()
    assert np.array_equal(ccnt, st_o[:, 1].astype(ccnt.dtype))
    ids, dists = st.rerank(10)
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))


@pytest.mark.parametrize("stride", [0, 128])
@pytest.mark.parametrize("fixture,ragged", [("small_f32", False), ("small_u8", False), ("small_u8", True), ("small_deep", True),
                                            ("small_i8", False)])
def test_pqdist_streaming_form_matches_oracle(request, libbang, fixture, ragged, stride):
    """bang_k_pqdist_stream (the launch the K2-alone roofline figure is measured on): same canonical float order, bit for bit,
    with ragged neighbour counts (0..64) and more queries than one sweep of the grid's waves handles (the ping-pong loop)."""
    import ctypes as C
    from bang_amd import binding as B
    from bang_amd.binding import IterState
    from oracle import oracle as O
    ix, q, _, _ = request.getfixturevalue(fixture)
    orc = O.Oracle(ix)
    Q = 700
    qq = np.ascontiguousarray(np.tile(q, ((Q + q.shape[0] - 1) // q.shape[0], 1))[:Q])
    st = IterState(ix, qq, 16, ragged=ragged)
    rng = np.random.default_rng(11)
    cnt = rng.integers(0, 65, Q).astype(np.uint32)
    cnt[:3] = (64, 0, 1)
    nb = np.zeros((Q, B.NBR_STRIDE), np.uint32)
    for i in range(Q):
        nb[i, :cnt[i]] = rng.choice(ix.N, cnt[i], replace=False)
    st.d_nbrs.upload(nb)
    st.d_cnt.upload(cnt)
    st.d_dist.zero()
    p = st.params()
    if stride:               # padded code table: rows 128 bytes apart (a row never leaves its line; rows start dword aligned)
        padded = np.zeros((ix.N, stride), np.uint8)
        padded[:, : ix.m] = ix.codes
        padded[:, ix.m:] = 0xA5                                # (whatever lies behind a row must not matter)
        d_pad = B.DeviceBuffer.from_numpy(padded, slack=256)
        p.d_codes, p.code_stride = d_pad.ptr, stride
    B._check(B.lib().bang_k_pqdist_stream(C.byref(p), None), "bang_k_pqdist_stream")
    B.sync()
    _, _, dist = st.nbrs()
    for i in list(range(0, Q, 37)) + [0, 1, 2, Q - 1]:
        lut = orc.lut_build(qq[i])
        want = orc.pqdist(lut, nb[i, :cnt[i]])
        assert np.array_equal(dist[i, :cnt[i]].view(np.uint32), want.view(np.uint32)), i
        assert not dist[i, cnt[i]:].any()
