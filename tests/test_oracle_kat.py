"""Known-answer tests that pin the CPU oracle (oracle/bang_oracle.c).

The reference ships no tests or golden vectors and cannot be built here (CUDA-only), so parity is UNPINNED BY
THE REFERENCE; these answers are derived by hand / by independent numpy arithmetic from the reference source
text (file:line cited per test; paths relative to /root/reference/BANG_Base/)."""
import numpy as np
import pytest

from oracle import oracle as O
from bang_amd.formats import Index, pack_graph

# hashFn1_d / hashFn2_d, bang_search.cu:1168-1189 -- values worked out from the formulas
HASH_KAT = {0: (122339, 65628), 1: (264707, 246870), 64: (223679, 62863), 12345: (46935, 275367),
            123742: (8978, 293991), 178757270: (147764, 3393), 999999999: (343906, 294772),
            4294967295: (366040, 298638)}


def test_hash_known_answers():
    for x, (h1, h2) in HASH_KAT.items():
        assert (O.hash1(x), O.hash2(x)) == (h1, h2)


def test_hash_matches_python_bigint_formula():
    rng = np.random.default_rng(1)
    for x in rng.integers(0, 2**32, 200, dtype=np.uint64):
        x = int(x)
        for basis, prime, fn in ((0xcbf29ce4, 0x01000193, O.hash1), (0x84222325, 0x1B3, O.hash2)):
            h = basis
            for i in range(4):
                h = ((h ^ ((x >> (8 * i)) & 0xff)) * prime) % 2**64      # uint64 wrap-around
            assert fn(x) == h % 399887


def test_bloom_constants():
    assert O.BF_ENTRIES == 399887 and O.BF_MEMORY == 399888        # bang_search.cu:48-50


def _toy_index(D=2, m=1, N=4, R=2, dtype="float", chunk_off=None, pivots=None, centroid=None, vectors=None):
    npd = {"float": np.float32, "uint8": np.uint8, "int8": np.int8}[dtype]
    vectors = np.zeros((N, D), npd) if vectors is None else vectors.astype(npd)
    deg = np.full(N, R, np.uint32)
    adj = (np.arange(N)[:, None] + 1 + np.arange(R)[None, :]) % N
    graph = pack_graph(vectors, deg, adj.astype(np.uint32))
    pivots = np.zeros((256, D), np.float32) if pivots is None else pivots
    centroid = np.zeros(D, np.float32) if centroid is None else centroid
    chunk_off = np.array([0, D], np.uint32) if chunk_off is None else chunk_off
    codes = np.zeros((N, m), np.uint8)
    return Index(dtype, N, D, R, m, 0, graph, codes, pivots, centroid, chunk_off)


def test_lut_toy_table():
    """populate_pqDist_par :1083-1130: LUT[c][k] = sum_j (P[k][j] - (q[j] - cen[j]))^2 over the chunk's dims."""
    piv = np.zeros((256, 2), np.float32)
    piv[0] = (1.0, 2.0)
    piv[1] = (-3.0, 0.5)
    piv[255] = (10.0, -10.0)
    ix = _toy_index(pivots=piv, centroid=np.array([0.5, -1.0], np.float32))
    lut = O.Oracle(ix).lut_build(np.array([2.0, 3.0], np.float32))
    # q - cen = (1.5, 4.0)
    assert lut[0, 0] == np.float32((1.0 - 1.5) ** 2 + (2.0 - 4.0) ** 2)       # 4.25
    assert lut[0, 1] == np.float32((-3.0 - 1.5) ** 2 + (0.5 - 4.0) ** 2)      # 32.5
    assert lut[0, 255] == np.float32((10 - 1.5) ** 2 + (-10 - 4.0) ** 2)      # 268.25
    assert lut[0, 7] == np.float32(1.5 ** 2 + 4.0 ** 2)                        # zero pivot


def test_lut_uses_fused_multiply_add_in_ascending_dim_order():
    """CANON: acc = fmaf(diff, diff, acc) for j ascending from +0.0f (`+=` at :1126 under nvcc's default fmad)."""
    rng = np.random.default_rng(3)
    D, m = 12, 3
    piv = rng.normal(size=(256, D)).astype(np.float32) * 50
    cen = rng.normal(size=D).astype(np.float32)
    q = rng.normal(size=D).astype(np.float32) * 30
    ix = _toy_index(D=D, m=m, pivots=piv, centroid=cen, chunk_off=np.array([0, 5, 9, 12], np.uint32))
    lut = O.Oracle(ix).lut_build(q)
    import math
    for c, (a, b) in enumerate([(0, 5), (5, 9), (9, 12)]):
        for k in (0, 17, 255):
            acc = np.float32(0.0)
            for j in range(a, b):
                diff = np.float32(piv[k, j] - np.float32(q[j] - cen[j]))
                # exact fused multiply-add in float64 is exact here (24-bit * 24-bit product fits 53 bits)
                acc = np.float32(float(diff) * float(diff) + float(acc))
            assert lut[c, k] == acc


def test_lut_mips_pads_query_with_zero():
    """n_DimAdjust = 1 (:1099-1113): the query has D-1 values, the last coordinate is 0."""
    piv = np.zeros((256, 3), np.float32)
    piv[5] = (1, 1, 1)
    ix = _toy_index(D=3, pivots=piv, centroid=np.array([0, 0, 0.25], np.float32), chunk_off=np.array([0, 3], np.uint32))
    lut = O.Oracle(ix).lut_build(np.array([1.0, 1.0], np.float32), dim_adjust=1)
    assert lut[0, 5] == np.float32((1 - (0 - 0.25)) ** 2)


def _canonical_k2(lut, row):
    m = len(row)
    s = []
    for l in range(8):
        acc = np.float32(0.0)
        for c in range(l, m, 8):
            acc = np.float32(acc + lut[c, row[c]])
        s.append(acc)
    a = np.float32(np.float32(s[0] + s[1]) + np.float32(s[2] + s[3]))
    b = np.float32(np.float32(s[4] + s[5]) + np.float32(s[6] + s[7]))
    return np.float32(a + b)


@pytest.mark.parametrize("m", [1, 7, 8, 32, 70, 74])
def test_pqdist_canonical_float_order(m):
    """compute_neighborDist_par :1225-1239: 8 strided partial sums, then the CUB shfl-down tree
    ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7)).  Checked against an independent float32 numpy evaluation; values are
    chosen so that a different association changes the result."""
    rng = np.random.default_rng(m)
    N = 40
    lut = (rng.random((m, 256)).astype(np.float32) * np.float32(1e4)) ** 2
    codes = rng.integers(0, 256, (N, m), dtype=np.uint8)
    ix = _toy_index(N=N, m=m)
    ix.codes = codes
    orc = O.Oracle(ix)
    ids = np.arange(N, dtype=np.uint32)
    got = orc.pqdist(lut, ids)
    want = np.array([_canonical_k2(lut, codes[i]) for i in range(N)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if m >= 32:
        naive = np.array([np.sum(lut[np.arange(m), codes[i]], dtype=np.float64) for i in range(N)]).astype(np.float32)
        assert not np.array_equal(naive, want)       # the order is observable


def test_filter_snapshot_semantics_and_order():
    """neighbor_filtering_new :1140-1165 with CANON snapshot semantics."""
    bloom = np.zeros(O.BF_MEMORY, np.uint8)
    first = O.filter_ids(bloom, np.array([7, 3, 7, 99], np.uint32))
    assert list(first) == [7, 3, 7, 99]             # both copies of 7 pass: tested against the state at entry
    assert bloom[O.hash1(7)] and bloom[O.hash2(7)] and bloom[O.hash1(99)]
    second = O.filter_ids(bloom, np.array([99, 5, 3], np.uint32))
    assert list(second) == [5]
    assert bloom.sum() <= 2 * 4                      # two slots per distinct id


def test_filter_false_positive_needs_both_slots():
    bloom = np.zeros(O.BF_MEMORY, np.uint8)
    bloom[O.hash1(42)] = 1
    assert list(O.filter_ids(bloom.copy(), np.array([42], np.uint32))) == [42]   # only one slot set -> kept (:1157)
    bloom[O.hash2(42)] = 1
    assert list(O.filter_ids(bloom, np.array([42], np.uint32))) == []


def test_sort_is_stable():
    """compute_BestLSets_par_sort_msort :1553-1584 (lower_bound for the left run, upper_bound for the right)."""
    ids = np.array([10, 11, 12, 13, 14, 15], np.uint32)
    d = np.array([3.0, 1.0, 3.0, 1.0, 0.5, 3.0], np.float32)
    si, sd = O.sort_pairs(ids, d)
    assert list(si) == [14, 11, 13, 10, 12, 15] and list(sd) == [0.5, 1, 1, 3, 3, 3]


def test_merge_new_entries_precede_equal_old_entries():
    """compute_BestLSets_par_merge :1675-1680: pos_new = lower_bound(old)+i, pos_old = upper_bound(new)+k."""
    w = O.merge([1, 2], [1.0, 2.0], 1, [], [], [], 4, medoid=1, mark=2)          # iter 1: init (:1638-1649)
    assert list(w[0]) == [1, 2] and list(w[2]) == [1, 1]                          # medoid visited, mark visited
    w = O.merge([7, 8], [2.0, 5.0], 2, w[0], w[1], w[2], 4, medoid=1, mark=99)
    assert list(w[0]) == [1, 7, 2, 8] and list(w[1]) == [1.0, 2.0, 2.0, 5.0]     # 7 (new, 2.0) before 2 (old, 2.0)
    assert list(w[2]) == [1, 0, 1, 0]


def test_merge_truncates_to_L_and_stops_at_first_not_better():
    w = ([1, 2, 3, 4], [1.0, 2.0, 3.0, 4.0], [0, 0, 0, 0])
    # new sorted: 0.5 < worst(4.0), 4.0 >= worst -> nb stops at 1 even though 1.5 follows (:1653-1657)
    out = O.merge([9, 8, 7], [0.5, 4.0, 1.5][:3], 5, *w, 4, medoid=0, mark=9)
    assert list(out[0]) == [9, 1, 2, 3] and list(out[2]) == [1, 0, 0, 0]


def test_merge_fills_an_unfull_worklist_even_with_worse_entries():
    """nbrsBound = max(nbrsBound, min(L - size, n)) :1660."""
    out = O.merge([9, 8], [10.0, 11.0], 3, [1], [1.0], [1], 4, medoid=0, mark=5)
    assert list(out[0]) == [1, 9, 8]


def test_parent1_skips_medoid_first_minimum_wins():
    ok, p, mk = O.parent1([5, 6, 7, 8], [0.1, 0.7, 0.3, 0.3], medoid=5)          # :1491-1503
    assert ok and p == 7 and mk == 7
    ok, _, _ = O.parent1([5], [0.1], medoid=5)                                    # CANON: nothing eligible
    assert not ok


def test_parent2_branches():
    w_ids, w_d = [1, 2, 3], [1.0, 2.0, 3.0]
    # best new (0.5) beats the first unvisited entry (2.0) -> parent = new, mark = new (:1429-1432)
    ok, p, mk, vis = O.parent2([9], [0.5], w_ids, w_d, [1, 0, 0], medoid=0)
    assert ok and p == 9 and mk == 9 and list(vis) == [1, 0, 0]
    # otherwise the worklist entry is taken and marked visited (:1433-1436); equal distance -> worklist wins
    ok, p, mk, vis = O.parent2([9], [2.0], w_ids, w_d, [1, 0, 0], medoid=0, mark=77)
    assert ok and p == 2 and mk == 77 and list(vis) == [1, 1, 0]
    # corner case :1442-1446: everything visited, new neighbour better than the worst entry
    ok, p, mk, vis = O.parent2([9], [2.5], w_ids, w_d, [1, 1, 1], medoid=0)
    assert ok and p == 9 and mk == 9
    ok, _, _, _ = O.parent2([9], [3.0], w_ids, w_d, [1, 1, 1], medoid=0)
    assert not ok
    # no new neighbours at all: only the worklist branch can fire (dist = 3.402823E+38 :1406)
    ok, p, _, _ = O.parent2([], [], w_ids, w_d, [1, 1, 0], medoid=0)
    assert ok and p == 3


def test_exact_distance_integer_subtraction():
    """compute_L2Dist :1293-1296: uint8 operands are subtracted as ints (no wrap-around)."""
    vec = np.zeros((2, 4), np.uint8)
    vec[1] = (0, 255, 10, 3)
    ix = _toy_index(D=4, N=2, dtype="uint8", vectors=vec, chunk_off=np.array([0, 4], np.uint32))
    d = O.Oracle(ix).exact_dist(1, np.array([255, 0, 13, 3], np.uint8))
    assert d == 255 ** 2 + 255 ** 2 + 9


def test_topk_stable_and_padded():
    ids, d = O.topk([4, 5, 6, 7], [2.0, 1.0, 2.0, 1.0], 3)                        # :1330-1367 ties keep expansion order
    assert list(ids) == [5, 7, 4] and list(d) == [1, 1, 2]
    ids, d = O.topk([4], [2.0], 3)                                                # CANON tail
    assert ids[0] == 4 and ids[1] == 2**64 - 1 and d[2] == np.float32(3.402823E+38)


def test_recall_counts_distance_ties():
    """calculate_recall, test_driver.cpp:43-93."""
    gt = np.array([[1, 2, 3, 4]], np.uint32)
    gd = np.array([[0.1, 0.2, 0.2, 0.9]], np.float32)
    res = np.array([[1, 3]], np.uint64)
    assert O.recall(gt, gd, res, 2) == 100.0      # id 3 ties with id 2 at rank 2 -> counted
    assert O.recall(gt, None, res, 2) == 50.0
    assert O.recall(gt, gd, np.array([[9, 8]], np.uint64), 2) == 0.0


def test_whole_search_small_graph_terminates_and_is_deterministic(small_u8):
    ix, q, gt_i, gt_d = small_u8
    orc = O.Oracle(ix)
    a = orc.search(q, 10, 30, nthreads=1)
    b = orc.search(q, 10, 30, nthreads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ids, dists, st = orc.search(q, 10, 30, with_stats=True)
    assert (st[:, 0] <= 30 + 49).all() and (st[:, 1] <= 30 + 50).all()           # cap :950-956
    assert O.recall(gt_i, gt_d, ids, 10) > 85
    # returned distances are the exact squared L2 of the returned ids, ascending
    v = ix.vectors().astype(np.float64)
    for i in (0, 5, 17):
        want = ((v[ids[i].astype(np.int64)] - q[i].astype(np.float64)) ** 2).sum(axis=1)
        assert np.allclose(dists[:, i], want) and (np.diff(dists[:, i]) >= 0).all()
