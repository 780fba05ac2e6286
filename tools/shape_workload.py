"""Shape-only workloads (SURVEY 8(d) Tier B) for throughput at the BASELINE.json scales:

  sift1b_shape   : N = 1e9, uint8, D = 128, R = 64, m = 70 -- graph + vectors (388 GB) in HOST RAM, PQ codes (70 GB) in HBM
  deep100m_shape : N = 1e8, float32, D = 96, R = 64, m = 74 -- graph + vectors (64.4 GB) and codes (7.4 GB) in HBM

Uniform random vectors / codes, random sorted R-regular adjacency, N(0,1)-scaled pivots: recall is NOT meaningful, the
per-query work (L = 152, iteration cap L+49, ~64 neighbours per expansion) and every memory footprint are.  N is scaled
down automatically to fit 75 % of the memory the box lets this process use (and says so)."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))

SHAPES = {
    "sift1b_shape": dict(N=1_000_000_000, D=128, dtype="uint8", R=64, m=70, graph="host"),
    "deep100m_shape": dict(N=100_000_000, D=96, dtype="float", R=64, m=74, graph="device"),
}


def _lib():
    so = os.path.join(HERE, "_build", "libshape_fill.so")
    src = os.path.join(HERE, "shape_fill.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        if os.environ.get("BANG_NO_BUILD"):          # a profiled child must never start a compiler (bench.py build_everything)
            raise RuntimeError(f"{so} is missing or stale and BANG_NO_BUILD is set")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    lib.shape_alloc.restype = C.c_void_p
    lib.shape_alloc.argtypes = [C.c_size_t]
    lib.shape_free.argtypes = [C.c_void_p, C.c_size_t]
    lib.shape_map_shared.restype = C.c_void_p
    lib.shape_map_shared.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
    lib.shape_fill_graph.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
    lib.shape_fill_bytes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    lib.shape_fill_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
    return lib


class ShapeSource(C.Structure):
    """shape_source of shape_fill.c: the context of shape_entry_source (the generator as an entry source of a streamed load)."""
    _fields_ = [("N", C.c_uint64), ("seed", C.c_uint64), ("vec_bytes", C.c_uint32), ("R", C.c_uint32),
                ("nthreads", C.c_int32), ("vec_float", C.c_int32)]


class LazyGraph:
    """Stands in for the [N][entry] graph image of a STREAMED shape index: rows are regenerated on demand (every node has its own
    generator state), which is all the result checks need (`graph[ids, :vec_bytes]`)."""

    def __init__(self, lib, src: ShapeSource, entry: int):
        self._lib, self._src, self._entry = lib, src, entry
        self.shape = (int(src.N), entry)

    def __getitem__(self, key):
        rows, cols = key if isinstance(key, tuple) else (key, slice(None))
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64))
        out = np.empty((rows.size, self._entry), np.uint8)
        s = self._src
        for k, r in enumerate(rows):
            self._lib.shape_fill_range(out[k].ctypes.data, s.N, int(r), 1, s.vec_bytes, s.R, s.seed, 1, s.vec_float)
        return out[:, cols]


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def usable_host_bytes() -> int:
    avail = 0
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            avail = int(line.split()[1]) * 1024
    try:
        m = open("/sys/fs/cgroup/memory.max").read().strip()
        if m != "max":
            used = int(open("/sys/fs/cgroup/memory.current").read())
            avail = min(avail, int(m) - used)
    except Exception:
        pass
    return avail


PULL_ROW_BYTES = 256       # host-graph placement, pull mode: the engine keeps the adjacency lists a second time as 256-byte rows
HOST_RESERVE = 40 << 30    # streamed load: host memory left to the process beside the pull rows (the engine itself insists on 8 GB; Python, torch,
                           # the pinned staging buffers of the load and the CPU-baseline leg's small index take the rest).  Round 3 used 75 % of
                           # the box -- 0.94e9 rows on a 320 GB box, whose 256 GB for N = 1e9 do fit


def stream_rows_budget():
    """pull rows a streamed load may pin in host memory on this box, and the bytes a 1e9-row index would be short of"""
    return max(0, usable_host_bytes() - HOST_RESERVE) // PULL_ROW_BYTES


CODE_STRIDE = {"sift1b_shape": 128, "deep100m_shape": 128}   # device-generated code tables: rows padded to their own 128-byte line


def code_stride(name) -> int:
    s = int(os.environ.get("BANG_SHAPE_CODE_STRIDE", "-1"))
    return CODE_STRIDE[name] if s < 0 else (s or SHAPES[name]["m"])


def plan_n(name, dev, n_override=0, reserve_rows=True, stream=False, shared_bytes=0):
    """N a shape workload will get on this box (after scaling to the host / HBM memory budget), without building anything.
    shared_bytes: free space of the directory the ranks of a node share (tmpfs) -- the ONE pull-rows file (and, not streamed, the
    graph image) must fit it: a mapping of a tmpfs file that outgrows the file system dies with SIGBUS, not with an error."""
    import torch
    sh = SHAPES[name]
    isz = 4 if sh["dtype"] == "float" else 1
    entry = sh["D"] * isz + 4 + 4 * sh["R"]
    N = n_override or int(os.environ.get("BANG_SHAPE_N", "0")) or sh["N"]
    if sh["graph"] == "host" and stream:
        free, _ = torch.cuda.mem_get_info(dev)
        N = min(N, stream_rows_budget(), (free - (28 << 30)) // (sh["D"] * isz + code_stride(name)))
    elif sh["graph"] == "host":
        N = min(N, int(usable_host_bytes() * 0.75) // (entry + (PULL_ROW_BYTES if reserve_rows else 0)))
    else:
        free, _ = torch.cuda.mem_get_info(dev)
        N = min(N, (int(free * 0.85) - (8 << 30)) // (entry + code_stride(name)), int(usable_host_bytes() * 0.75) // entry)
    if shared_bytes and sh["graph"] == "host":
        per_node = (PULL_ROW_BYTES if (stream or reserve_rows) else 0) + (0 if stream else entry)
        if per_node:
            N = min(N, int(shared_bytes * 0.9) // per_node)
    return int(N)


class ShapeIndex:
    """Duck-types bang_amd.formats.Index for Engine.load_index (graph is a numpy view of the mmap'ed image)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def entry_len(self):
        return self.D * (4 if self.dtype == "float" else 1) + 4 + 4 * self.R


class _RawDev:
    """a device allocation made with hipExtMallocWithFlags, visible to torch through __cuda_array_interface__ (never freed: experiments)"""

    def __init__(self, nbytes, flags):
        hip = C.CDLL("libamdhip64.so")
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(flags))
        if rc != 0 or not p.value:
            raise MemoryError(f"hipExtMallocWithFlags({nbytes}, {flags:#x}) -> {rc}")
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (p.value, False), "version": 2}


def _raw_device_bytes(nbytes, flags, dev):
    import torch
    return torch.as_tensor(_RawDev(nbytes, flags), device=dev)


def make(name, dev, n_override=0, Q=10_000, seed=20240711, log=print, host_codes=False, shared=None, reserve_rows=True,
         stream=False, planned=False):
    """host_codes=True: the PQ codes are generated in HOST memory too (ix.codes, uploaded by bang_load) so that the CPU oracle
    can run on the index (parity tests at > 4 GiB offsets); default: straight on the device, host copy absent.
    shared=(path, is_creator, barrier): the graph image lives in ONE mapping of the file `path` shared by every rank of the node
    (the creator fills it, the others map it after `barrier()`); N is derived from the memory of the node, not of a rank."""
    import torch
    from bang_amd.synth import chunk_offsets
    sh = dict(SHAPES[name])
    isz = 4 if sh["dtype"] == "float" else 1
    D, R, m = sh["D"], sh["R"], sh["m"]
    cs = m if host_codes else code_stride(name)              # bytes between the rows of the device-generated code table
    entry = D * isz + 4 + 4 * R
    N = n_override or int(os.environ.get("BANG_SHAPE_N", "0")) or sh["N"]
    lib = _lib()
    ncpu = usable_cpus()
    note = ""
    stream = bool(stream and sh["graph"] == "host" and not host_codes)
    if n_override and (planned or shared is not None):
        pass                                   # N was planned once for the node (plan_n on rank 0) and handed to every rank: final.  Ranks that
                                               # re-scaled it from their own momentary memory figures could end up with different indices
    elif stream:
        # STREAMED: the graph image never exists -- the engine pulls the generator's entries through in chunks (vectors -> HBM,
        # adjacency lists -> 256-byte pull rows in host memory).  Host budget: the rows; HBM budget: codes + vectors + 28 GB.
        free, _total = torch.cuda.mem_get_info(dev)
        n_host, n_hbm = stream_rows_budget(), (free - (28 << 30)) // (D * isz + cs)
        N2 = min(N, n_host, n_hbm)
        if N2 < N:
            short_host = max(0, N - n_host) * PULL_ROW_BYTES / 1e9
            short_hbm = max(0, N - n_hbm) * (D * isz + cs) / 1e9
            note = (f" (N scaled {N} -> {N2}: the box is {short_host:.1f} GB of host memory short of the {N * PULL_ROW_BYTES / 1e9:.0f} GB of pull rows "
                    f"[{usable_host_bytes() / 1e9:.0f} GB usable, {HOST_RESERVE >> 30} GB kept free] and {short_hbm:.1f} GB of HBM short of vectors + codes + 28 GB)")
            N = N2
    elif sh["graph"] == "host":
        budget = int(usable_host_bytes() * 0.75)
        per_node = entry + (PULL_ROW_BYTES if reserve_rows else 0)
        if N * per_node > budget:
            N2 = budget // per_node
            note = (f" (N scaled {N} -> {N2}: host memory budget {budget / 2**30:.0f} GiB for {entry}-byte graph entries"
                    + (f" + {PULL_ROW_BYTES}-byte pull rows" if reserve_rows else "") + ")")
            N = N2
    else:
        free, _total = torch.cuda.mem_get_info(dev)
        budget = int(free * 0.85) - (8 << 30)
        if N * (entry + cs) > budget:
            N2 = budget // (entry + cs)
            note = f" (N scaled {N} -> {N2}: HBM budget {budget / 2**30:.0f} GiB)"
            N = N2
        host_budget = int(usable_host_bytes() * 0.75)
        if N * entry > host_budget:
            N2 = host_budget // entry
            note += f" (N scaled to {N2}: host staging budget)"
            N = N2
    t0 = time.time()
    gbytes = N * entry
    ptr, graph, source = None, None, None
    if stream:
        source = ShapeSource(N, seed, D * isz, R, ncpu, 1 if sh["dtype"] == "float" else 0)
        graph = LazyGraph(lib, source, entry)
        gbytes = 0
    elif shared is not None:
        path, creator, barrier = shared
        if creator:
            ptr = lib.shape_map_shared(path.encode(), gbytes, 1)
            if not ptr:
                raise MemoryError(f"cannot create the shared mapping {path} ({gbytes} bytes)")
            lib.shape_fill_graph(ptr, N, D * isz, R, seed, ncpu, 1 if sh["dtype"] == "float" else 0)
            barrier()
        else:
            barrier()
            ptr = lib.shape_map_shared(path.encode(), gbytes, 0)
            if not ptr:
                raise MemoryError(f"cannot map the shared graph {path}")
    else:
        ptr = lib.shape_alloc(gbytes)
        if not ptr:
            raise MemoryError(f"cannot map {gbytes} bytes")
        lib.shape_fill_graph(ptr, N, D * isz, R, seed, ncpu, 1 if sh["dtype"] == "float" else 0)   # floats: uniform in [-1, 1)
    if not stream:
        graph = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(gbytes,)).reshape(N, entry)
        log(f"[shape] graph image {gbytes / 2**30:.1f} GiB filled in {time.time() - t0:.1f}s with {ncpu} threads{note}")
    t0 = time.time()
    codes_host, cptr = None, None
    if host_codes:
        cptr = lib.shape_alloc(N * m)
        if not cptr:
            raise MemoryError(f"cannot map {N * m} bytes")
        lib.shape_fill_bytes(cptr, N * m, seed + 7, ncpu)
        codes_host = np.ctypeslib.as_array(C.cast(cptr, C.POINTER(C.c_uint8)), shape=(N * m,)).reshape(N, m)
        codes = None
        log(f"[shape] {N * m / 2**30:.1f} GiB of PQ codes generated in host memory in {time.time() - t0:.1f}s")
    else:                               # PQ codes straight on the device
        if os.environ.get("SHAPE_CODES_MEMFLAGS"):      # experiment (tools/dev): the code table in memory of another type (hipExtMallocWithFlags)
            codes = _raw_device_bytes(N * cs + 256, int(os.environ["SHAPE_CODES_MEMFLAGS"], 0), dev)
        else:
            codes = torch.empty(N * cs + 256, dtype=torch.uint8, device=dev)   # (uniform random bytes: the padding behind a row too)
        step = 1 << 28
        g = torch.Generator(device=dev)
        g.manual_seed(seed + 7)
        for a in range(0, N * cs, step):
            b = min(N * cs, a + step)
            codes[a:b] = torch.randint(0, 256, (b - a,), dtype=torch.uint8, device=dev, generator=g)
        codes[N * cs:] = 0
        torch.cuda.synchronize()
        log(f"[shape] {N * cs / 2**30:.1f} GiB of PQ codes (rows {cs} bytes apart) generated on the device in {time.time() - t0:.1f}s")
    rng = np.random.default_rng(seed + 1)
    scale = 40.0 if sh["dtype"] == "uint8" else 0.5
    pivots = (rng.standard_normal((256, D)) * scale).astype(np.float32)
    centroid = (np.full(D, 127.5) if sh["dtype"] == "uint8" else np.zeros(D)).astype(np.float32)
    if sh["dtype"] == "uint8":
        queries = rng.integers(0, 256, (Q, D), dtype=np.uint8)
    else:
        queries = (rng.random((Q, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    ix = ShapeIndex(dtype=sh["dtype"], N=N, D=D, R=R, m=m, medoid=int(N // 2), graph=graph,
                    codes=codes_host if host_codes else np.zeros((1, m), np.uint8), pivots=pivots, centroid=centroid,
                    chunk_off=chunk_offsets(D, m), _ptr=ptr, _bytes=gbytes, _codes=codes, _lib=lib, _cptr=cptr, _cbytes=N * m,
                    entry_source=((lib.shape_entry_source, source) if stream else None), code_stride=(0 if host_codes else cs))
    # (the driver's record keeps ~120 characters of a string: N, layout and placement first, prose behind)
    head = f"{name} N={N} {sh['dtype']} D={D} R={R} m={m} Q={Q} shape-only"
    if stream:
        name_s = (f"{head} STREAMED: rows {N * PULL_ROW_BYTES / 1e9:.0f} GB host RAM (pull rows), vectors {N * D * isz / 1e9:.0f} GB + codes "
                  f"{N * cs / 1e9:.0f} GB ({cs} B/row) HBM{note}")
    else:
        name_s = (f"{head}: graph+vectors {gbytes / 1e9:.0f} GB in {'host RAM' if sh['graph'] == 'host' else 'HBM'}, "
                  f"codes {N * cs / 1e9:.0f} GB ({cs} B/row) HBM{note}")
    return ix, queries, None, None, (codes.data_ptr() if codes is not None else None), name_s, sh["graph"]


def release(ix):
    """Unmap the host images of a shape index (the numpy views of it must not be used afterwards)."""
    if getattr(ix, "_ptr", None):
        ix.graph = None
        ix._lib.shape_free(ix._ptr, ix._bytes)
        ix._ptr = None
    if getattr(ix, "_cptr", None):
        ix.codes = None
        ix._lib.shape_free(ix._cptr, ix._cbytes)
        ix._cptr = None
    ix._codes = None
