"""ctypes binding of include/bang_c.h (engine level + kernel level)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.environ.get("BANG_AMD_LIB") or os.path.join(_PKG, "lib", "libbang.so")    # (override: A/B runs of experimental builds)

U8, I8, F32 = 0, 1, 2
DIST_L2, DIST_MIPS = 0, 1
GRAPH_HOST, GRAPH_DEVICE, GRAPH_AUTO = 0, 1, 2
DTYPE_CODE = {"uint8": U8, "int8": I8, "float": F32}
NP_DTYPE = {"uint8": np.uint8, "int8": np.int8, "float": np.float32}

BF_WORDS = 12512
NBR_STRIDE = 72
STAGE_STRIDE = 65
EXTRA_ITERS = 50
NO_PARENT = 0xFFFFFFFF
IDLE_PARENT = 0xFFFFFFFE


class BangError(RuntimeError):
    pass


class IterParams(C.Structure):
    _fields_ = [
        ("Q", C.c_uint32), ("R", C.c_uint32), ("m", C.c_uint32), ("L", C.c_uint32), ("medoid", C.c_uint32),
        ("iter", C.c_uint32), ("psz", C.c_uint32), ("mp", C.c_uint32), ("first", C.c_uint32), ("max_wgs", C.c_uint32), ("d_qmap", C.c_void_p), ("n_all", C.c_uint32),
        ("d_stage", C.c_void_p), ("d_seed", C.c_void_p), ("d_codes", C.c_void_p), ("d_pivots_packed", C.c_void_p),
        ("d_qc", C.c_void_p), ("d_lut", C.c_void_p), ("d_graph", C.c_void_p), ("entry_len", C.c_uint64),
        ("vec_bytes", C.c_uint32),
        ("d_bloom", C.c_void_p), ("d_nbrs", C.c_void_p), ("d_dist", C.c_void_p), ("d_cnt", C.c_void_p),
        ("d_wl_ids", C.c_void_p), ("d_wl_dist", C.c_void_p), ("d_wl_vis", C.c_void_p), ("d_wl_cnt", C.c_void_p),
        ("d_mark", C.c_void_p), ("d_parents", C.c_void_p), ("d_cand_ids", C.c_void_p), ("d_cand_row", C.c_void_p),
        ("d_cand_cnt", C.c_void_p), ("d_active", C.c_void_p), ("d_qstats", C.c_void_p),
        ("d_done_count", C.c_void_p), ("h_done_flag", C.c_void_p), ("d_ktime", C.c_void_p), ("h_parents", C.c_void_p), ("done_value", C.c_uint32),
        ("pq_nhi", C.c_uint32), ("code_stride", C.c_uint32),
    ]


class SearchParams(C.Structure):
    """bang_search_params (include/bang_c.h)."""
    _fields_ = [
        ("Q", C.c_uint32), ("R", C.c_uint32), ("m", C.c_uint32), ("L", C.c_uint32), ("medoid", C.c_uint32), ("cap_iter", C.c_uint32),
        ("psz", C.c_uint32), ("mp", C.c_uint32), ("pq_nhi", C.c_uint32), ("max_wgs", C.c_uint32), ("max_waves", C.c_uint32),
        ("d_seed", C.c_void_p), ("d_codes", C.c_void_p), ("d_pivots_packed", C.c_void_p), ("d_qc", C.c_void_p),
        ("d_graph", C.c_void_p), ("entry_len", C.c_uint64), ("vec_bytes", C.c_uint32), ("row_layout", C.c_uint32),
        ("d_bloom", C.c_void_p), ("d_cand_ids", C.c_void_p), ("d_cand_cnt", C.c_void_p), ("d_qstats", C.c_void_p),
        ("d_qiters", C.c_void_p), ("d_next_query", C.c_void_p), ("d_ktime", C.c_void_p),
        ("d_rows", C.c_void_p), ("d_ctl", C.c_void_p), ("h_done", C.c_void_p), ("h_parents", C.c_void_p), ("h_pub_q", C.c_void_p),
        ("h_pub_c", C.c_void_p), ("d_abort", C.c_void_p), ("ship_vectors", C.c_uint32), ("nctx", C.c_uint32),
        ("group_waves", C.c_uint32), ("d_prof", C.c_void_p), ("code_stride", C.c_uint32), ("n_rows_hbm", C.c_uint32), ("d_rows_hbm", C.c_void_p),
        ("d_row_slices", C.c_void_p), ("n_slices", C.c_uint32), ("slice_rows", C.c_uint32), ("go_timeout_ticks", C.c_uint64), ("d_qskip", C.c_void_p),
        ("n_nodes", C.c_uint32), ("summ_iters", C.c_uint32), ("spec_rows", C.c_uint32),
        ("rr_queries", C.c_void_p), ("rr_vec_base", C.c_void_p), ("rr_vec_stride", C.c_uint64), ("rr_ids_out", C.c_void_p), ("rr_dists_out", C.c_void_p),
        ("rr_dtype", C.c_uint32), ("rr_D", C.c_uint32), ("rr_k", C.c_uint32), ("rr_q0", C.c_uint32), ("rr_Q_total", C.c_uint32),
    ]


class IndexDesc(C.Structure):
    _fields_ = [("medoid", C.c_uint64), ("entry_len", C.c_uint64), ("D", C.c_uint32), ("R", C.c_uint32),
                ("N", C.c_uint32), ("m", C.c_uint32), ("graph", C.c_void_p), ("codes", C.c_void_p),
                ("d_codes", C.c_void_p), ("pivots", C.c_void_p), ("centroid", C.c_void_p), ("chunk_off", C.c_void_p),
                ("code_stride", C.c_uint32), ("vectors_ready", C.c_uint32), ("d_vectors", C.c_void_p), ("rows_hash", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("wall_ms", C.c_double), ("iterations", C.c_uint64), ("dist_evals", C.c_uint64),
                ("fetched", C.c_uint64), ("candidates", C.c_uint64), ("front_launches", C.c_uint64),
                ("front_ms", C.c_double), ("back_ms", C.c_double), ("rerank_ms", C.c_double),
                ("walker_ms", C.c_double), ("sync_ms", C.c_double), ("enqueue_ms", C.c_double), ("front_busy_ms", C.c_double),
                ("persistent", C.c_uint64), ("h2d_bytes", C.c_uint64), ("vectors_on_device", C.c_uint64),
                ("graph_mode", C.c_uint64), ("lanes", C.c_uint64), ("walker_threads", C.c_uint64), ("wg_queries", C.c_uint64),
                ("workgroups", C.c_uint64), ("hops_p50", C.c_uint64), ("hops_p99", C.c_uint64), ("hops_max", C.c_uint64),
                ("search_kernel", C.c_uint64), ("pacing_groups", C.c_uint64), ("graph_pull", C.c_uint64), ("pulled_bytes", C.c_uint64),
                ("rows_in_hbm", C.c_uint64), ("code_stride", C.c_uint64), ("filter_loads_skipped", C.c_uint64), ("rows_from_peer", C.c_uint64), ("rows_from_own_hbm", C.c_uint64), ("walker_rows", C.c_uint64), ("rerank_fused", C.c_uint64)]


ENTRY_SOURCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p)     # bang_entry_source


def lib_path() -> str:
    return _LIB


def _newer(target: str, sources) -> bool:
    return (not os.path.exists(target)) or any(os.path.getmtime(x) > os.path.getmtime(target) for x in sources)


def build(force: bool = False) -> str:
    """Compile libbang.so + bang_search + bang_search_multi for gfx950 (hipcc cross-compiles without a GPU).  What is stale is decided the
    way the Makefile decides it: the library against ITS sources (kernels, engine, headers), each harness against its own source and the
    library -- a harness edited after the last build must not make the library look stale (a profiled child, BANG_NO_BUILD, only needs the
    library and must never start a compiler)."""
    csrc = os.path.join(_PKG, "csrc")
    harness = {"bang_search": "test_driver.cpp", "bang_search_multi": "test_driver_multi.cpp"}
    lib_srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f not in harness.values()]
    lib_srcs += [os.path.join(_PKG, "..", "include", f) for f in ("bang.h", "bang_c.h")]
    lib_stale = _newer(_LIB, lib_srcs)
    bins_stale = any(_newer(os.path.join(_PKG, "bin", exe), [os.path.join(csrc, src)] + ([_LIB] if os.path.exists(_LIB) else []))
                     for exe, src in harness.items())
    if os.environ.get("BANG_AMD_LIB"):
        return _LIB
    if os.environ.get("BANG_NO_BUILD"):              # a profiled child must never start a compiler (bench.py build_everything)
        if lib_stale:
            raise RuntimeError(f"{_LIB} is missing or stale and BANG_NO_BUILD is set")
        return _LIB
    if force or lib_stale or bins_stale:
        subprocess.check_call(["make", "-C", _PKG, "-s", "-j4"])
    return _LIB


_lib = None


def lib():
    """Load libbang.so.  Fails loudly when it is missing: there is no fallback implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise BangError(f"{_LIB} not found: run `make -C {_PKG}` (or __graft_entry__.build()) first; "
                            "bang_amd has no Python/CPU fallback")
        _lib = C.CDLL(_LIB, mode=C.RTLD_GLOBAL)
        _lib.bang_last_error.restype = C.c_char_p
        for name in ("bang_device_count",):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise BangError(f"{what} failed (code {rc}): {lib().bang_last_error().decode(errors='replace')}")


def device_count() -> int:
    return int(lib().bang_device_count())


def _vp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """Raw device allocation (bang_dev_*)."""

    def __init__(self, nbytes: int, zero: bool = True):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(lib().bang_dev_malloc(C.byref(p), C.c_size_t(self.nbytes)), "bang_dev_malloc")
        self.ptr = p.value
        if zero and self.nbytes:
            _check(lib().bang_dev_memset(C.c_void_p(self.ptr), 0, C.c_size_t(self.nbytes)), "bang_dev_memset")

    @classmethod
    def from_numpy(cls, a: np.ndarray, slack: int = 0) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes + slack, zero=slack > 0)
        if a.nbytes:
            _check(lib().bang_dev_h2d(C.c_void_p(b.ptr), _vp(a), C.c_size_t(a.nbytes)), "bang_dev_h2d")
        return b

    def upload(self, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        _check(lib().bang_dev_h2d(C.c_void_p(self.ptr + offset), _vp(a), C.c_size_t(a.nbytes)), "bang_dev_h2d")

    def download(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(lib().bang_dev_d2h(_vp(out), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)), "bang_dev_d2h")
        return out

    def zero(self):
        _check(lib().bang_dev_memset(C.c_void_p(self.ptr), 0, C.c_size_t(self.nbytes)), "bang_dev_memset")

    def free(self):
        if self.ptr:
            lib().bang_dev_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync():
    _check(lib().bang_dev_sync(), "bang_dev_sync")


class Engine:
    """Engine-level API: same call order as BANGSearch<T> (bang.h) / test_driver.cpp."""

    def __init__(self, dtype: str, **options):
        self.dtype = dtype
        self._h = C.c_void_p()
        _check(lib().bang_create(C.c_int(DTYPE_CODE[dtype]), C.byref(self._h)), "bang_create")
        self._keep = []
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, key: str, value: int):
        _check(lib().bang_set_option(self._h, key.encode(), C.c_long(int(value))), f"bang_set_option({key})")

    def load(self, prefix: str):
        _check(lib().bang_load_e(self._h, prefix.encode()), "bang_load")

    def load_index(self, ix, d_codes: int | None = None, code_stride: int = 0):
        """Load from a formats.Index held in memory (graph stays referenced, not copied, in host mode).  d_codes: the PQ codes are
        already on the device, rows `code_stride` bytes apart (0 = m)."""
        graph = np.ascontiguousarray(ix.graph, dtype=np.uint8)
        codes = np.ascontiguousarray(ix.codes, dtype=np.uint8)
        pivots = np.ascontiguousarray(ix.pivots, dtype=np.float32)
        centroid = np.ascontiguousarray(ix.centroid, dtype=np.float32)
        chunk_off = np.ascontiguousarray(ix.chunk_off, dtype=np.uint32)
        self._keep = [graph, codes, pivots, centroid, chunk_off]
        d = IndexDesc(ix.medoid, ix.entry_len, ix.D, ix.R, ix.N, ix.m, _vp(graph).value, _vp(codes).value,
                      d_codes, _vp(pivots).value, _vp(centroid).value, _vp(chunk_off).value, code_stride)
        _check(lib().bang_load_mem_e(self._h, C.byref(d)), "bang_load_mem")

    def load_stream(self, ix, source, ctx=None, d_codes: int | None = None, code_stride: int = 0, d_vectors: int | None = None):
        """Streamed load (bang_load_stream_e): `source` is a C function pointer -- or a Python callable
        (first, count, dst_address) -> 0 -- that writes `count` graph entries in the reference layout for the nodes from `first`;
        ix supplies everything but the graph (medoid, entry_len, D, R, N, m, codes, pivots, centroid, chunk_off)."""
        codes = None if ix.codes is None else np.ascontiguousarray(ix.codes, dtype=np.uint8)
        pivots = np.ascontiguousarray(ix.pivots, dtype=np.float32)
        centroid = np.ascontiguousarray(ix.centroid, dtype=np.float32)
        chunk_off = np.ascontiguousarray(ix.chunk_off, dtype=np.uint32)
        if not isinstance(source, (ENTRY_SOURCE, C._CFuncPtr)):
            py = source
            source = ENTRY_SOURCE(lambda _ctx, first, count, dst: int(py(first, count, dst)))
        self._keep = [codes, pivots, centroid, chunk_off, source]
        d = IndexDesc(ix.medoid, ix.entry_len, ix.D, ix.R, ix.N, ix.m, None, None if codes is None else _vp(codes).value,
                      d_codes, _vp(pivots).value, _vp(centroid).value, _vp(chunk_off).value, code_stride, 0, d_vectors, 0)
        fn = lib().bang_load_stream_e
        fn.argtypes = [C.c_void_p, C.POINTER(IndexDesc), C.c_void_p, C.c_void_p]
        _check(fn(self._h, C.byref(d), C.cast(source, C.c_void_p), ctx), "bang_load_stream")

    def load_shared(self, ix, d_vectors: int, rows_hash: int, d_codes: int | None = None, code_stride: int = 0):
        """bang_load_shared_e: this rank did not read the index -- the vectors are already in its device buffer `d_vectors` (received
        from the loading rank), the adjacency rows in the node's rows file (BANG_PULL_ROWS_DIR) whose hash the loading rank reported."""
        codes = None if ix.codes is None else np.ascontiguousarray(ix.codes, dtype=np.uint8)
        pivots = np.ascontiguousarray(ix.pivots, dtype=np.float32)
        centroid = np.ascontiguousarray(ix.centroid, dtype=np.float32)
        chunk_off = np.ascontiguousarray(ix.chunk_off, dtype=np.uint32)
        self._keep = [codes, pivots, centroid, chunk_off]
        d = IndexDesc(ix.medoid, ix.entry_len, ix.D, ix.R, ix.N, ix.m, None, None if codes is None else _vp(codes).value,
                      d_codes, _vp(pivots).value, _vp(centroid).value, _vp(chunk_off).value, code_stride, 1, d_vectors, rows_hash)
        fn = lib().bang_load_shared_e
        fn.argtypes = [C.c_void_p, C.POINTER(IndexDesc)]
        _check(fn(self._h, C.byref(d)), "bang_load_shared")

    def rows_hash(self) -> int:
        h = C.c_uint64()
        _check(lib().bang_get_rows_hash(self._h, C.byref(h)), "bang_get_rows_hash")
        return int(h.value)

    # ---- peer rows (include/bang_c.h): the node's adjacency rows in the node's spare HBM
    def rows_capacity(self) -> int:
        n = C.c_uint64()
        _check(lib().bang_rows_capacity_e(self._h, C.byref(n)), "bang_rows_capacity")
        return int(n.value)

    def rows_slice(self, first_row: int, rows: int):
        _check(lib().bang_rows_slice_e(self._h, C.c_uint64(first_row), C.c_uint64(rows)), "bang_rows_slice")

    def rows_export(self):
        """-> (64-byte IPC handle (all zero: nothing to share), first row, rows)"""
        h = (C.c_ubyte * 64)()
        first, rows = C.c_uint64(), C.c_uint64()
        _check(lib().bang_rows_export_e(self._h, h, C.byref(first), C.byref(rows)), "bang_rows_export")
        return bytes(h), int(first.value), int(rows.value)

    def rows_import(self, slot: int, n_slots: int, slice_rows: int, handle: bytes | None, rows: int = 0):
        """rows: what the exporter's rows_export() reported (a peer slice shorter than the slot is refused)."""
        buf = (C.c_ubyte * 64).from_buffer_copy(handle) if handle is not None else None
        _check(lib().bang_rows_import_e(self._h, C.c_uint32(slot), C.c_uint32(n_slots), C.c_uint64(slice_rows), C.c_uint64(rows), buf), "bang_rows_import")

    def rows_close_peers(self):
        _check(lib().bang_rows_close_peers_e(self._h), "bang_rows_close_peers")

    def set_searchparams(self, recall: int, worklist_length: int, distfn: int = DIST_L2):
        _check(lib().bang_set_searchparams_e(self._h, recall, worklist_length, distfn), "bang_set_searchparams")
        self.k, self.L = recall, worklist_length

    def alloc(self, num_queries: int):
        _check(lib().bang_alloc_e(self._h, num_queries), "bang_alloc")

    def init(self, num_queries: int):
        _check(lib().bang_init_e(self._h, num_queries), "bang_init")

    def query(self, queries: np.ndarray, ids_out: np.ndarray | None = None, dists_out: np.ndarray | None = None):
        q = np.ascontiguousarray(queries, dtype=NP_DTYPE[self.dtype])
        Q = q.shape[0]
        ids = ids_out if ids_out is not None else np.empty((Q, self.k), dtype=np.uint64)
        dists = dists_out if dists_out is not None else np.empty((self.k, Q), dtype=np.float32)
        _check(lib().bang_query_e(self._h, _vp(q), Q, _vp(ids), _vp(dists)), "bang_query")
        return ids, dists

    def query_dev(self, queries: np.ndarray, d_ids: int, d_dists: int = 0):
        """bang_query_dev_e: the result ids [Q][k] u64 (and, if d_dists != 0, the distances [k][Q] f32) are written to DEVICE buffers of
        the caller (raw device addresses, e.g. torch.Tensor.data_ptr()); nothing returns to the host."""
        q = np.ascontiguousarray(queries, dtype=NP_DTYPE[self.dtype])
        fn = lib().bang_query_dev_e
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _check(fn(self._h, _vp(q), q.shape[0], C.c_void_p(d_ids), C.c_void_p(d_dists) if d_dists else None), "bang_query_dev")

    def stats(self) -> dict:
        s = Stats()
        _check(lib().bang_get_stats(self._h, C.byref(s)), "bang_get_stats")
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    def query_counters(self, Q: int) -> np.ndarray:
        """[Q][4] per-query {iterations (search kernel only, else 0), candidates, dist_evals, fetched} of the last query -- the
        column order of the oracle's per-query statistics."""
        cols = [np.zeros(Q, np.uint32) for _ in range(4)]
        _check(lib().bang_get_query_counters(self._h, _vp(cols[2]), _vp(cols[3]), _vp(cols[1]), _vp(cols[0])), "bang_get_query_counters")
        return np.stack(cols, axis=1).astype(np.int64)

    def candidate_log(self, Q: int, L: int):
        """(ids [Q][L + 50] u32, counts [Q]): the nodes every query of the last batch expanded, in expansion order."""
        ids = np.zeros((Q, L + EXTRA_ITERS), np.uint32)
        cnt = np.zeros(Q, np.uint32)
        _check(lib().bang_get_candidate_log(self._h, _vp(ids), C.c_uint32(ids.shape[1]), _vp(cnt), C.c_uint32(Q)), "bang_get_candidate_log")
        return ids, cnt

    def free(self):
        _check(lib().bang_free_e(self._h), "bang_free")

    def unload(self):
        _check(lib().bang_unload_e(self._h), "bang_unload")
        self._keep = []

    def close(self):
        if self._h:
            lib().bang_destroy(self._h)
            self._h = None
            self._keep = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pq_layout(chunk_off: np.ndarray, D: int, m: int):
    psz, mp = C.c_uint32(0), C.c_uint32(0)
    co = np.ascontiguousarray(chunk_off, dtype=np.uint32)
    _check(lib().bang_pq_layout(_vp(co), D, m, C.byref(psz), C.byref(mp)), "bang_pq_layout")
    return psz.value, mp.value


def pack_pivots_ragged(pivots: np.ndarray, chunk_off: np.ndarray, D: int, m: int, mp: int):
    """(nhi, table) for the exact-size 2-float layout, or (0, None) if the chunk sizes are not 2,..,2,1,..,1."""
    pv = np.ascontiguousarray(pivots, dtype=np.float32)
    co = np.ascontiguousarray(chunk_off, dtype=np.uint32)
    nhi, nfl = C.c_uint32(0), C.c_uint64(0)
    _check(lib().bang_pack_pivots_ragged(None, _vp(co), D, m, mp, C.byref(nhi), None, C.byref(nfl)), "bang_pack_pivots_ragged")
    if nhi.value == 0:
        return 0, None
    out = np.zeros(nfl.value, np.float32)
    _check(lib().bang_pack_pivots_ragged(_vp(pv), _vp(co), D, m, mp, C.byref(nhi), _vp(out), C.byref(nfl)), "bang_pack_pivots_ragged")
    return int(nhi.value), out


def pack_pivots(pivots: np.ndarray, chunk_off: np.ndarray, D: int, m: int, psz: int, mp: int) -> np.ndarray:
    out = np.empty((mp, 256, psz), dtype=np.float32)
    pv = np.ascontiguousarray(pivots, dtype=np.float32)
    co = np.ascontiguousarray(chunk_off, dtype=np.uint32)
    _check(lib().bang_pack_pivots(_vp(pv), _vp(co), D, m, psz, mp, _vp(out)), "bang_pack_pivots")
    return out


class IterState:
    """Device-side state of Q queries for driving the KERNEL-level entries one by one (tests, bench).

    ``use_lut=True`` selects the LUT path (K1 builds [Q][m][256]; K2 gathers from it);
    otherwise the pivot-stationary path is used when the index layout allows it."""

    def __init__(self, ix, queries: np.ndarray, L: int, use_lut: bool = False, dim_adjust: int = 0,
                 device_graph: bool = False, ragged: bool = False):
        self.ix, self.L = ix, L
        self.Q = Q = queries.shape[0]
        self.dim_adjust = dim_adjust
        self.dtype_code = DTYPE_CODE[ix.dtype]
        q = np.ascontiguousarray(queries, dtype=NP_DTYPE[ix.dtype])
        rows = L + EXTRA_ITERS
        self.rows = rows
        self.d_queries = DeviceBuffer.from_numpy(q, slack=16)
        self.d_codes = DeviceBuffer.from_numpy(ix.codes, slack=256)
        self.d_centroid = DeviceBuffer.from_numpy(ix.centroid.astype(np.float32))
        self.d_chunk_off = DeviceBuffer.from_numpy(ix.chunk_off.astype(np.uint32))
        self.d_pivots_T = DeviceBuffer.from_numpy(np.ascontiguousarray(ix.pivots.T, dtype=np.float32))
        psz, mp = pq_layout(ix.chunk_off, ix.D, ix.m)
        if use_lut:
            psz, mp = 0, ix.m
        self.psz, self.mp = psz, mp
        self.d_pivots_packed = self.d_qc = self.d_lut = None
        self.pq_nhi = 0
        if psz == 2 and ragged:      # exact-size table (chunks 2,..,2,1,..,1 dims wide), if the layout has that form
            self.pq_nhi, table = pack_pivots_ragged(ix.pivots, ix.chunk_off, ix.D, ix.m, mp)
            if self.pq_nhi:
                self.d_pivots_packed = DeviceBuffer.from_numpy(table)
        if psz:
            if not self.pq_nhi:
                self.d_pivots_packed = DeviceBuffer.from_numpy(pack_pivots(ix.pivots, ix.chunk_off, ix.D, ix.m, psz, mp))
            self.d_qc = DeviceBuffer(Q * mp * psz * 4)
            _check(lib().bang_k_center_queries(C.c_void_p(self.d_queries.ptr), self.dtype_code,
                                               C.c_void_p(self.d_centroid.ptr), C.c_void_p(self.d_chunk_off.ptr),
                                               C.c_void_p(self.d_qc.ptr), Q, ix.D, ix.m, mp, psz, dim_adjust, None),
                   "bang_k_center_queries")
        else:
            self.d_lut = DeviceBuffer(Q * ix.m * 256 * 4)
            self.lut_build()
        adj = ix.adjacency()[ix.medoid][: int(ix.degrees()[ix.medoid])]
        seed = np.zeros(2 + 65, dtype=np.uint32)
        seed[0] = 1 + len(adj)
        seed[1] = ix.medoid
        seed[2:2 + len(adj)] = adj
        self.d_seed = DeviceBuffer.from_numpy(seed)
        self.d_graph = DeviceBuffer.from_numpy(ix.graph, slack=256) if device_graph else None
        self.d_stage = DeviceBuffer(Q * STAGE_STRIDE * 4)
        self.d_bloom = DeviceBuffer(Q * BF_WORDS * 4)
        self.d_nbrs = DeviceBuffer(Q * NBR_STRIDE * 4)
        self.d_dist = DeviceBuffer(Q * NBR_STRIDE * 4)
        self.d_cnt = DeviceBuffer(Q * 4)
        self.d_wl_ids = DeviceBuffer(Q * L * 4)
        self.d_wl_dist = DeviceBuffer(Q * L * 4)
        self.d_wl_vis = DeviceBuffer(Q * L)
        self.d_wl_cnt = DeviceBuffer(Q * 4)
        self.d_mark = DeviceBuffer(Q * 4)
        self.d_parents = DeviceBuffer(Q * 4)
        self.d_cand_ids = DeviceBuffer(Q * rows * 4)
        self.d_cand_row = DeviceBuffer(Q * rows * 4)
        self.d_cand_cnt = DeviceBuffer(Q * 4)
        self.d_active = DeviceBuffer(4 * (rows + 2))
        self.d_qstats = DeviceBuffer(Q * 8)
        self.iter = 1
        self.first = 1
        self.reset()

    def lut_build(self):
        ix = self.ix
        _check(lib().bang_k_lut_build(C.c_void_p(self.d_pivots_T.ptr), C.c_void_p(self.d_queries.ptr), self.dtype_code,
                                      C.c_void_p(self.d_centroid.ptr), C.c_void_p(self.d_chunk_off.ptr),
                                      C.c_void_p(self.d_lut.ptr), self.Q, ix.D, ix.m, self.dim_adjust, None),
               "bang_k_lut_build")
        sync()

    def reset(self):
        self.d_bloom.zero()
        self.d_qstats.zero()
        self.d_active.zero()
        _check(lib().bang_k_init_state(self.Q, self.ix.medoid, self.rows, C.c_void_p(self.d_cand_ids.ptr),
                                       C.c_void_p(self.d_cand_row.ptr), C.c_void_p(self.d_cand_cnt.ptr),
                                       C.c_void_p(self.d_wl_cnt.ptr), C.c_void_p(self.d_mark.ptr),
                                       C.c_void_p(self.d_parents.ptr), C.c_void_p(self.d_cnt.ptr), None),
               "bang_k_init_state")
        sync()
        self.iter, self.first = 1, 1

    def params(self) -> IterParams:
        ix = self.ix
        p = IterParams()
        p.Q, p.R, p.m, p.L, p.medoid, p.iter = self.Q, ix.R, ix.m, self.L, ix.medoid, self.iter
        p.psz, p.mp, p.first = self.psz, self.mp, self.first
        p.pq_nhi = self.pq_nhi
        p.d_stage, p.d_seed, p.d_codes = self.d_stage.ptr, self.d_seed.ptr, self.d_codes.ptr
        p.d_pivots_packed = self.d_pivots_packed.ptr if self.d_pivots_packed else None
        p.d_qc = self.d_qc.ptr if self.d_qc else None
        p.d_lut = self.d_lut.ptr if self.d_lut else None
        p.d_graph = self.d_graph.ptr if self.d_graph else None
        p.entry_len = ix.entry_len
        p.vec_bytes = ix.D * np.dtype(NP_DTYPE[ix.dtype]).itemsize
        p.d_bloom, p.d_nbrs, p.d_dist, p.d_cnt = self.d_bloom.ptr, self.d_nbrs.ptr, self.d_dist.ptr, self.d_cnt.ptr
        p.d_wl_ids, p.d_wl_dist, p.d_wl_vis, p.d_wl_cnt = (self.d_wl_ids.ptr, self.d_wl_dist.ptr, self.d_wl_vis.ptr,
                                                          self.d_wl_cnt.ptr)
        p.d_mark, p.d_parents = self.d_mark.ptr, self.d_parents.ptr
        p.d_cand_ids, p.d_cand_row, p.d_cand_cnt = self.d_cand_ids.ptr, self.d_cand_row.ptr, self.d_cand_cnt.ptr
        p.d_active = None
        p.d_qstats = self.d_qstats.ptr
        return p

    def run(self, entry: str):
        """entry in {"front", "back", "filter", "pqdist", "parent"}"""
        p = self.params()
        _check(getattr(lib(), "bang_k_" + entry)(C.byref(p), None), "bang_k_" + entry)
        sync()

    def run_search(self):
        """The whole search loop in ONE launch of the query-resident search kernel (graph resident in HBM: needs
        device_graph=True).  Fills the candidate log; returns the per-query iteration counts."""
        assert self.d_graph is not None and self.psz != 0
        ix = self.ix
        sp = SearchParams()
        sp.Q, sp.R, sp.m, sp.L, sp.medoid, sp.cap_iter = self.Q, ix.R, ix.m, self.L, ix.medoid, self.L + EXTRA_ITERS - 1
        sp.psz, sp.mp, sp.pq_nhi = self.psz, self.mp, self.pq_nhi
        sp.d_seed, sp.d_codes, sp.d_pivots_packed, sp.d_qc = self.d_seed.ptr, self.d_codes.ptr, self.d_pivots_packed.ptr, self.d_qc.ptr
        sp.d_graph, sp.entry_len = self.d_graph.ptr, ix.entry_len
        sp.vec_bytes = ix.D * np.dtype(NP_DTYPE[ix.dtype]).itemsize
        sp.d_bloom, sp.d_cand_ids, sp.d_cand_cnt, sp.d_qstats = self.d_bloom.ptr, self.d_cand_ids.ptr, self.d_cand_cnt.ptr, self.d_qstats.ptr
        d_iters = DeviceBuffer(self.Q * 4)
        d_next = DeviceBuffer(64)
        sp.d_qiters, sp.d_next_query = d_iters.ptr, d_next.ptr
        _check(lib().bang_k_search(C.byref(sp), None), "bang_k_search")
        sync()
        return d_iters.download(np.uint32, (self.Q,))

    def stage(self, lists):
        """Upload per-query adjacency lists (what the host walker stages every iteration)."""
        st = np.zeros((self.Q, STAGE_STRIDE), dtype=np.uint32)
        for q, l in enumerate(lists):
            st[q, 0] = len(l)
            st[q, 1:1 + len(l)] = l
        self.d_stage.upload(st)

    # --- downloads ---
    def nbrs(self):
        cnt = self.d_cnt.download(np.uint32, (self.Q,))
        ids = self.d_nbrs.download(np.uint32, (self.Q, NBR_STRIDE))
        dist = self.d_dist.download(np.float32, (self.Q, NBR_STRIDE))
        return cnt, ids, dist

    def worklist(self):
        n = self.d_wl_cnt.download(np.uint32, (self.Q,))
        return (n, self.d_wl_ids.download(np.uint32, (self.Q, self.L)), self.d_wl_dist.download(np.float32, (self.Q, self.L)),
                self.d_wl_vis.download(np.uint8, (self.Q, self.L)))

    def parents(self):
        return self.d_parents.download(np.uint32, (self.Q,)), self.d_mark.download(np.uint32, (self.Q,))

    def candidates(self):
        return (self.d_cand_cnt.download(np.uint32, (self.Q,)), self.d_cand_ids.download(np.uint32, (self.Q, self.rows)),
                self.d_cand_row.download(np.uint32, (self.Q, self.rows)))

    def lut(self):
        return self.d_lut.download(np.float32, (self.Q, self.ix.m, 256))

    def rerank(self, k: int):
        """K6+K7 with the vectors taken straight from a device copy of the graph."""
        if self.d_graph is None:
            self.d_graph = DeviceBuffer.from_numpy(self.ix.graph, slack=256)
        ix = self.ix
        d_ids = DeviceBuffer(self.Q * k * 8)
        d_dists = DeviceBuffer(self.Q * k * 4)
        medoid_vec = DeviceBuffer.from_numpy(np.ascontiguousarray(ix.graph[ix.medoid]), slack=16)
        _check(lib().bang_k_rerank(C.c_void_p(self.d_graph.ptr), C.c_uint64(ix.entry_len), C.c_void_p(medoid_vec.ptr),
                                   C.c_void_p(self.d_queries.ptr), self.dtype_code, C.c_void_p(self.d_cand_ids.ptr),
                                   None, C.c_void_p(self.d_cand_cnt.ptr), self.rows, self.Q, ix.D, k, self.dim_adjust,
                                   C.c_void_p(d_ids.ptr), C.c_void_p(d_dists.ptr), None), "bang_k_rerank")
        sync()
        return d_ids.download(np.uint64, (self.Q, k)), d_dists.download(np.float32, (k, self.Q))
