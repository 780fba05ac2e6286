"""ctypes binding of the CPU oracle (oracle/bang_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product.  PARITY UNPINNED BY THE REFERENCE (see bang_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

BF_ENTRIES = 399887
BF_MEMORY = (BF_ENTRIES & 0xFFFFFFFC) + 4
DTYPE_CODE = {"uint8": 0, "int8": 1, "float": 2}
NP_DTYPE = {"uint8": np.uint8, "int8": np.int8, "float": np.float32}


class OrcIndex(C.Structure):
    _fields_ = [("medoid", C.c_uint64), ("entry_len", C.c_uint64), ("D", C.c_uint32), ("R", C.c_uint32),
                ("N", C.c_uint32), ("m", C.c_uint32), ("dtype", C.c_int32), ("graph", C.c_void_p),
                ("codes", C.c_void_p), ("pivots_T", C.c_void_p), ("centroid", C.c_void_p),
                ("chunk_off", C.c_void_p)]


class OrcQStats(C.Structure):
    _fields_ = [("iterations", C.c_uint32), ("candidates", C.c_uint32), ("dist_evals", C.c_uint64),
                ("fetched", C.c_uint64)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "bang_oracle.c")):
        if os.environ.get("BANG_NO_BUILD"):          # a profiled child must never start a compiler (bench.py build_everything)
            raise RuntimeError(f"{_LIB_PATH} is missing or stale and BANG_NO_BUILD is set")
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_hash1.restype = C.c_uint32
        _lib.orc_hash1.argtypes = [C.c_uint32]
        _lib.orc_hash2.restype = C.c_uint32
        _lib.orc_hash2.argtypes = [C.c_uint32]
        _lib.orc_filter.restype = C.c_uint32
        _lib.orc_merge.restype = C.c_uint32
        _lib.orc_parent1.restype = C.c_int
        _lib.orc_parent2.restype = C.c_int
        _lib.orc_exact_dist.restype = C.c_float
        _lib.orc_search_batch.restype = C.c_int
        _lib.orc_recall.restype = C.c_double
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Holds an index image (formats.Index) in the layout bang_load builds."""

    def __init__(self, ix):
        self.ix = ix
        self.graph = np.ascontiguousarray(ix.graph, dtype=np.uint8)
        self.codes = np.ascontiguousarray(ix.codes, dtype=np.uint8)
        self.pivots_T = np.ascontiguousarray(ix.pivots.T, dtype=np.float32)  # [D][256], bang_search.cu:281-285
        self.centroid = np.ascontiguousarray(ix.centroid, dtype=np.float32)
        self.chunk_off = np.ascontiguousarray(ix.chunk_off, dtype=np.uint32)
        self.view = OrcIndex(ix.medoid, ix.entry_len, ix.D, ix.R, ix.N, ix.m, DTYPE_CODE[ix.dtype],
                             _p(self.graph).value, _p(self.codes).value, _p(self.pivots_T).value,
                             _p(self.centroid).value, _p(self.chunk_off).value)

    # ---- stage functions -------------------------------------------------
    def lut_build(self, query: np.ndarray, dim_adjust: int = 0) -> np.ndarray:
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.ix.dtype])
        out = np.empty((self.ix.m, 256), dtype=np.float32)
        lib().orc_lut_build(C.byref(self.view), _p(q), C.c_int(dim_adjust), _p(out))
        return out

    def pqdist(self, lut: np.ndarray, ids: np.ndarray) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        out = np.empty(ids.shape[0], dtype=np.float32)
        lib().orc_pqdist(_p(lut), _p(self.codes), C.c_uint32(self.ix.m), _p(ids), C.c_uint32(ids.shape[0]), _p(out))
        return out

    def exact_dist(self, node: int, query: np.ndarray, dim_adjust: int = 0) -> float:
        q = np.ascontiguousarray(query, dtype=NP_DTYPE[self.ix.dtype])
        vec = np.ascontiguousarray(self.graph[node])
        return float(lib().orc_exact_dist(_p(vec), _p(q), C.c_uint32(self.ix.D), C.c_int(DTYPE_CODE[self.ix.dtype]),
                                          C.c_int(dim_adjust)))

    # ---- whole search -----------------------------------------------------
    def search(self, queries: np.ndarray, k: int, L: int, mips: bool = False, nthreads: int = 0,
               with_stats: bool = False):
        q = np.ascontiguousarray(queries, dtype=NP_DTYPE[self.ix.dtype])
        Q = q.shape[0]
        ids = np.empty((Q, k), dtype=np.uint64)
        dists = np.empty((k, Q), dtype=np.float32)
        stats = (OrcQStats * Q)() if with_stats else None
        rc = lib().orc_search_batch(C.byref(self.view), _p(q), C.c_uint32(Q), C.c_uint32(k), C.c_uint32(L),
                                    C.c_int(1 if mips else 0), _p(ids), _p(dists),
                                    stats if with_stats else None, C.c_int(nthreads))
        if rc != 0:
            raise RuntimeError(f"orc_search_batch failed: {rc}")
        if with_stats:
            st = np.array([(s.iterations, s.candidates, s.dist_evals, s.fetched) for s in stats], dtype=np.int64)
            return ids, dists, st
        return ids, dists


# ---- free stage functions (no index needed) ---------------------------------
def hash1(x: int) -> int:
    return int(lib().orc_hash1(C.c_uint32(x)))


def hash2(x: int) -> int:
    return int(lib().orc_hash2(C.c_uint32(x)))


def filter_ids(bloom: np.ndarray, ids: np.ndarray) -> np.ndarray:
    assert bloom.dtype == np.uint8 and bloom.shape[0] == BF_MEMORY
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.empty_like(ids)
    n = lib().orc_filter(_p(bloom), _p(ids), C.c_uint32(ids.shape[0]), _p(out))
    return out[:n].copy()


def sort_pairs(ids: np.ndarray, dist: np.ndarray):
    ids = np.array(ids, dtype=np.uint32)
    dist = np.array(dist, dtype=np.float32)
    lib().orc_sort_pairs(_p(ids), _p(dist), C.c_uint32(ids.shape[0]))
    return ids, dist


def merge(s_ids, s_dist, it: int, w_ids, w_dist, w_vis, L: int, medoid: int, mark: int):
    s_ids = np.ascontiguousarray(s_ids, dtype=np.uint32)
    s_dist = np.ascontiguousarray(s_dist, dtype=np.float32)
    wi = np.zeros(512, dtype=np.uint32)
    wd = np.zeros(512, dtype=np.float32)
    wv = np.zeros(512, dtype=np.uint8)
    n = len(w_ids)
    wi[:n], wd[:n], wv[:n] = w_ids, w_dist, w_vis
    new_n = lib().orc_merge(_p(s_ids), _p(s_dist), C.c_uint32(len(s_ids)), C.c_uint32(it), _p(wi), _p(wd), _p(wv),
                            C.c_uint32(n), C.c_uint32(L), C.c_uint32(medoid), C.c_uint32(mark))
    return wi[:new_n].copy(), wd[:new_n].copy(), wv[:new_n].copy()


def parent1(s_ids, s_dist, medoid: int):
    s_ids = np.ascontiguousarray(s_ids, dtype=np.uint32)
    s_dist = np.ascontiguousarray(s_dist, dtype=np.float32)
    p, mk = C.c_uint32(0), C.c_uint32(0)
    ok = lib().orc_parent1(_p(s_ids), _p(s_dist), C.c_uint32(len(s_ids)), C.c_uint32(medoid), C.byref(p), C.byref(mk))
    return bool(ok), p.value, mk.value


def parent2(s_ids, s_dist, w_ids, w_dist, w_vis, medoid: int, mark: int = 0):
    s_ids = np.ascontiguousarray(s_ids, dtype=np.uint32)
    s_dist = np.ascontiguousarray(s_dist, dtype=np.float32)
    w_ids = np.ascontiguousarray(w_ids, dtype=np.uint32)
    w_dist = np.ascontiguousarray(w_dist, dtype=np.float32)
    wv = np.array(w_vis, dtype=np.uint8)
    p, mk = C.c_uint32(0), C.c_uint32(mark)
    ok = lib().orc_parent2(_p(s_ids), _p(s_dist), C.c_uint32(len(s_ids)), _p(w_ids), _p(w_dist), _p(wv),
                           C.c_uint32(len(w_ids)), C.c_uint32(medoid), C.byref(p), C.byref(mk))
    return bool(ok), p.value, mk.value, wv


def topk(cand_ids, cand_dist, k: int):
    cand_ids = np.ascontiguousarray(cand_ids, dtype=np.uint32)
    cand_dist = np.ascontiguousarray(cand_dist, dtype=np.float32)
    ids = np.empty(k, dtype=np.uint64)
    d = np.empty(k, dtype=np.float32)
    lib().orc_topk(_p(cand_ids), _p(cand_dist), C.c_uint32(len(cand_ids)), C.c_uint32(k), _p(ids), _p(d))
    return ids, d


def recall(gt_ids: np.ndarray, gt_dists, results: np.ndarray, recall_at: int) -> float:
    gt_ids = np.ascontiguousarray(gt_ids, dtype=np.uint32)
    res = np.ascontiguousarray(results, dtype=np.uint64)
    gd = None if gt_dists is None else np.ascontiguousarray(gt_dists, dtype=np.float32)
    return float(lib().orc_recall(C.c_uint32(gt_ids.shape[0]), _p(gt_ids), _p(gd) if gd is not None else None,
                                  C.c_uint32(gt_ids.shape[1]), _p(res), C.c_uint32(res.shape[1]),
                                  C.c_uint32(recall_at)))


def num_threads() -> int:
    return int(lib().orc_num_threads())
