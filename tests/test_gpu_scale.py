"""BASELINE.json configs at (or near) their stated sizes on the GPU.

* configs[0] (SIFT10K-sized plumbing case through the harness CLI, BANG_Base/ReadMe.pdf p.3): N = 10 000, float32, D = 128,
  R = 64, m = 32, Q = 100, `bang_search <prefix> <query> <gt> 100 10 float l2 auto` -- the printed recall of every L on the
  sweep grid equals the oracle's, and the ids obtained through bang_load on the same FILES equal the oracle's bit for bit.
* configs[2]/[3] layouts (DEEP100M: float32 x 96 in 74 chunks; SIFT1B: uint8 x 128 in 70 chunks) at an N whose PQ-code
  table AND graph are larger than 4 GiB, so that every 64-bit offset of the path is exercised -- `id * m`
  (bang_search.cu:1232), `parent * entry_len` in the device graph and in the host walker (:796-810), `id * vec_bytes` of the
  resident vectors -- against the oracle on a small batch, bit for bit, in every placement."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config0_size_through_the_cli(libbang, tmp_path):
    import bang_amd
    from bang_amd import formats, synth
    from oracle import oracle as O
    N, D, R, m, Q, k = 10_000, 128, 64, 32, 100, 10
    ix, q, gt_i, gt_d = synth.make_index(N, D, "float", R, m, Q, K=k, n_clusters=64, seed=20240711, device="cpu", pq_iters=6)
    prefix = str(tmp_path / "sift10k_index")
    formats.write_index(prefix, ix)
    formats.write_bin(str(tmp_path / "siftsmall_query.bin"), q)
    formats.write_truthset(str(tmp_path / "sift10k_groundtruth.bin"), gt_i, gt_d)
    exe = os.path.join(os.path.dirname(os.path.dirname(bang_amd.lib_path())), "bin", "bang_search")
    orc = O.Oracle(ix)
    for graph in ("host", "device", "auto"):
        out = subprocess.run([exe, prefix, str(tmp_path / "siftsmall_query.bin"), str(tmp_path / "sift10k_groundtruth.bin"),
                              str(Q), str(k), "float", "l2", "auto"], capture_output=True, text=True, timeout=1200,
                             env=dict(os.environ, BANG_GRAPH=graph))
        assert out.returncode == 0, out.stderr[-2000:]
        rows = [l.split("\t") for l in out.stdout.splitlines() if l[:1].isdigit() and l.count("\t") == 3]
        Ls = sorted({int(r[0]) for r in rows})
        assert Ls == list(range(k, 513, 12))                                   # test_driver.cpp:376-417
        for L in Ls:
            mine = [r[3].strip() for r in rows if int(r[0]) == L]
            assert len(mine) == 5 and len(set(mine)) == 1                      # 5 runs per L, deterministic
            if graph == "host" or L in (10, 58, 154, 502):
                ids_o, _ = orc.search(q, k, L)
                assert mine[0] == "%.2f" % O.recall(gt_i, gt_d, ids_o, k), (graph, L)
        assert float([r[3] for r in rows if int(r[0]) == 154][0]) >= 90.0      # a meaningful index
    for graph in (bang_amd.GRAPH_HOST, bang_amd.GRAPH_DEVICE):
        with bang_amd.Engine("float", graph=graph) as e:
            e.load(prefix)
            for L in (10, 70, 200):
                ids_o, dists_o = orc.search(q, k, L)
                e.set_searchparams(k, L)
                e.alloc(Q)
                e.init(Q)
                ids, dists = e.query(q)
                e.free()
                assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32))
            e.unload()


def test_load_rejects_wrong_dtype_and_late_options(libbang, tmp_path):
    """bang_load refuses an index whose metadata contradicts the engine's element type (the reference would read vectors and
    adjacency lists at the wrong offsets); placement options cannot change after bang_load, loop options after bang_alloc."""
    import bang_amd
    from bang_amd import formats, synth
    ix, q, _, _ = synth.make_index(800, 32, "uint8", 16, 8, 8, K=5, n_clusters=4, seed=5, device="cpu", pq_iters=2)
    prefix = str(tmp_path / "u8")
    formats.write_index(prefix, ix)
    for wrong in ("int8", "float"):
        with bang_amd.Engine(wrong) as e:
            with pytest.raises(bang_amd.BangError, match="dtype|entry length"):
                e.load(prefix)
    with bang_amd.Engine("uint8", graph=0) as e:
        e.load(prefix)
        for key in ("graph", "device", "pq", "vectors"):
            with pytest.raises(bang_amd.BangError, match="before bang_load"):
                e.set_option(key, 1)
        e.set_searchparams(5, 20)
        e.alloc(8)
        for key in ("lanes", "threads", "persistent", "stage_zero_copy"):
            with pytest.raises(bang_amd.BangError, match="before bang_alloc"):
                e.set_option(key, 1)
        e.init(8)
        e.query(q)
        st = e.stats()
        assert st["graph_mode"] == 0 and st["lanes"] >= 1 and st["hops_max"] >= st["hops_p50"] >= 1
        assert st["graph_pull"] == 1 and st["walker_threads"] == 0   # the default of the host placement when the vectors are resident


@pytest.mark.parametrize("shape,N", [("sift1b_shape", 64_000_000), ("deep100m_shape", 60_000_000)])
def test_offsets_beyond_4gib_match_oracle(libbang, shape, N):
    import torch
    import bang_amd
    from oracle import oracle as O
    from tools import shape_workload as SW
    Q, k, L = 48, 10, 40
    ix, q, _, _, _, name, _ = SW.make(shape, torch.device("cuda", 0), n_override=N, Q=Q, log=lambda *a: None, host_codes=True)
    try:
        assert ix.N == N, f"box too small for this test: {name}"
        assert ix.N * ix.m > 2**32 and ix.N * ix.entry_len > 2**32 and ix.N * (ix.entry_len - 4 - 4 * ix.R) > 2**32
        ids_o, dists_o, st_o = O.Oracle(ix).search(q, k, L, with_stats=True)
        assert ids_o.max() > 2**32 // ix.entry_len                      # results do live beyond the 4 GiB mark
        for opts in (dict(graph=0, pull=1), dict(graph=0, pull=1, walker=1), dict(graph=0, vectors=1, pull=0), dict(graph=0, vectors=0),
                     dict(graph=0, persistent=0), dict(graph=1)):
            with bang_amd.Engine(ix.dtype, **opts) as e:
                e.load_index(ix)
                e.set_searchparams(k, L)
                e.alloc(Q)
                e.init(Q)
                ids, dists = e.query(q)
                st = e.stats()
                e.free()
                e.unload()
            assert np.array_equal(ids, ids_o), opts
            assert np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), opts
            assert st["dist_evals"] == int(st_o[:, 2].sum()) and st["candidates"] == int(st_o[:, 1].sum())
    finally:
        SW.release(ix)


def test_streamed_shape_index_equals_the_resident_one(libbang):
    """The streamed load at a size whose pull rows exceed 4 GiB: the shape generator as the engine's entry source (no graph image in
    host memory) against the same index built as a resident image -- pulled, walked and in HBM.  Same ids, same distances."""
    import ctypes as C
    import torch
    import bang_amd
    from tools import shape_workload as SW
    N, Q, k, L = 20_000_000, 64, 10, 40
    dev = torch.device("cuda", 0)
    ix_s, q, _, _, d_codes_s, name_s, _ = SW.make("sift1b_shape", dev, n_override=N, Q=Q, log=lambda *a: None, stream=True)
    ix_r, q_r, _, _, d_codes_r, _, _ = SW.make("sift1b_shape", dev, n_override=N, Q=Q, log=lambda *a: None, stream=False)
    try:
        assert ix_s.N == N and ix_r.N == N and "STREAMED" in name_s and N * 256 > 2**32
        assert np.array_equal(q, q_r)
        assert np.array_equal(ix_s.graph[np.array([0, N // 2, N - 1])], ix_r.graph[[0, N // 2, N - 1]])   # one generator, two routes

        def run(load, **opts):
            with bang_amd.Engine(ix_s.dtype, **opts) as e:
                load(e)
                e.set_searchparams(k, L)
                e.alloc(Q)
                e.init(Q)
                ids, dists = e.query(q)
                st = e.stats()
            return ids, dists, st
        src = ix_s.entry_source
        ids, dists, st = run(lambda e: e.load_stream(ix_s, src[0], C.byref(src[1]), d_codes=d_codes_s), graph=0)
        assert st["graph_pull"] == 1 and st["pulled_bytes"] == 256 * (st["candidates"] - Q)
        assert (ids < N).all() and ids.max() > 2**32 // 256                 # results do live behind the 4 GiB mark of the rows
        for opts in (dict(graph=0, pull=1), dict(graph=0, pull=0), dict(graph=1)):
            ids2, dists2, st2 = run(lambda e: e.load_index(ix_r, d_codes=d_codes_r), **opts)
            assert np.array_equal(ids, ids2) and np.array_equal(dists.view(np.uint32), dists2.view(np.uint32)), opts
            assert st2["dist_evals"] == st["dist_evals"] and st2["candidates"] == st["candidates"]
    finally:
        SW.release(ix_s)
        SW.release(ix_r)
