"""Randomised index shapes on the GPU vs the oracle: odd dimensions, chunk counts that need padding or fall back to the
LUT path, small degree bounds with ragged lists, maximum worklist length, k = L, single-query batches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # N,    D,  dtype,   R,  m,  Q,  k,  L
    (900,   24, "uint8", 16, 6,  9,  3,  3),      # k == L, 4 dims per chunk, m padded 6 -> 16
    (1200,  50, "float", 64, 25, 7,  10, 64),     # 2 dims per chunk, m = 25 -> padded to 32, rows not dword aligned
    (800,   37, "int8",  32, 37, 5,  4,  40),     # 1 dim per chunk (psz 1), odd D (byte path of the re-rank)
    (700,   96, "float", 48, 12, 11, 10, 512),    # 8 dims per chunk (psz 8, m padded 12 -> 16), maximum L
    (600,   90, "uint8", 64, 5,  3,  5,  25),     # 18 dims per chunk -> LUT path (no LDS layout)
    (1500, 128, "uint8", 64, 64, 1,  10, 30),     # single query, m = 64 (2 dims per chunk, 16 dwords)
    (1000,  64, "float", 8,  16, 13, 8,  100),    # tiny degree bound: the worklist rarely fills
    (500,  100, "int8",  64, 100, 6, 10, 20),     # m = 100 -> psz 1, 25 dwords -> padded to 32
]


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}-D{c[1]}-{c[2]}-R{c[3]}-m{c[4]}-Q{c[5]}-k{c[6]}-L{c[7]}" for c in CASES])
@pytest.mark.parametrize("graph", [0, 1, "walker"])      # 0: host placement (pull mode where the layout allows), "walker": host, pull = 0
def test_random_config_matches_oracle(libbang, case, graph):
    import bang_amd
    from bang_amd import synth
    from oracle import oracle as O
    N, D, dtype, R, m, Q, k, L = case
    ix, q, _, _ = synth.make_index(N, D, dtype, R, m, Q, K=min(10, k), n_clusters=8, seed=1000 + N + D, pq_iters=2)
    ids_o, dists_o, st_o = O.Oracle(ix).search(q, k, L, with_stats=True)
    opts = dict(graph=0, pull=0) if graph == "walker" else dict(graph=graph)
    with bang_amd.Engine(dtype, **opts) as e:
        e.load_index(ix)
        e.set_searchparams(k, L)
        e.alloc(Q)
        e.init(Q)
        ids, dists = e.query(q)
        st = e.stats()
        e.free()
        e.unload()
    assert np.array_equal(ids, ids_o)
    assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
    assert st["dist_evals"] == int(st_o[:, 2].sum()) and st["candidates"] == int(st_o[:, 1].sum())
