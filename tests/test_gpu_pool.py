"""The K2 pool of the search kernel (csrc/bang_search.hip) -- an experiment that is compiled OUT of libbang.so (it measured
no faster) but stays bit-identical: the build with it (`lib_pool/libbang.so`, made by __graft_entry__.build()) runs here in a child
process (one process loads one libbang) against the oracle: ids, distances and per-query counters, both placements, batches smaller
than the wave slots (helpers from the start) and far larger (the drain)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bang-billion-scale-ann_amd", "lib_pool", "libbang.so")

CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["BANG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BANG_ROOT"], "bang-billion-scale-ann_amd"))
import bang_amd
from bang_amd import synth
from oracle import oracle as O
cases = 0
for (N, D, dt, m, seed) in ((4000, 128, "uint8", 70, 12), (3000, 96, "float", 74, 14)):
    ix, q, _, _ = synth.make_index(N, D, dt, 64, m, 64, K=10, n_clusters=16, seed=seed, pq_iters=3)
    orc = O.Oracle(ix)
    for graph in (0, 1):
        for L in (10, 64, 152):
            ids_o, dists_o, st_o = orc.search(q, 10, L, with_stats=True)
            with bang_amd.Engine(ix.dtype, graph=graph, search=1, pool=1) as e:
                e.load_index(ix); e.set_searchparams(10, L); e.alloc(q.shape[0])
                for _ in range(2):
                    e.init(q.shape[0])
                    ids, dists = e.query(q)
                    st = e.stats()
                    assert st["search_kernel"] == 1 and st["pool_jobs"] > 0, st
                    assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), (N, graph, L)
                    assert np.array_equal(e.query_counters(q.shape[0]), st_o), (N, graph, L)
                e.free(); e.unload()
            cases += 1
    if dt == "uint8":                                   # batch sizes: one query; fewer than waves; far more than the wave slots of two workgroups
        rng = np.random.default_rng(5)
        for Q in (1, 5, 700):
            qq = np.ascontiguousarray(np.concatenate([q] * (Q // q.shape[0] + 1))[:Q])
            qq = np.clip(qq.astype(np.int32) + rng.integers(-3, 4, qq.shape), 0, 255).astype(np.uint8)
            ids_o, dists_o, st_o = orc.search(qq, 10, 48, with_stats=True)
            if Q > 100: os.environ["BANG_SEARCH_MAX_WGS"] = "2"
            with bang_amd.Engine(ix.dtype, graph=1, search=1, pool=1) as e:
                e.load_index(ix); e.set_searchparams(10, 48); e.alloc(Q); e.init(Q)
                ids, dists = e.query(qq)
                assert np.array_equal(ids, ids_o) and np.array_equal(dists.view(np.uint32), dists_o.view(np.uint32)), Q
                assert np.array_equal(e.query_counters(Q), st_o) and e.stats()["pool_jobs"] > 0, Q
                e.free(); e.unload()
            os.environ.pop("BANG_SEARCH_MAX_WGS", None)
            cases += 1
print("POOL_OK", cases)
"""


def test_k2_pool_build_matches_oracle():
    if not os.path.exists(LIB):
        pytest.skip("lib_pool/libbang.so not built (make OUT=lib_pool EXTRA_CXXFLAGS=-DBANG_SEARCH_POOL=1)")
    env = dict(os.environ, BANG_AMD_LIB=LIB, BANG_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "POOL_OK 15" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
