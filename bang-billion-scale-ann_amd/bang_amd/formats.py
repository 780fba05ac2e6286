"""On-disk formats of the BANG_Base search path (readers + writers).

These are the files ``bang_load`` consumes (reference: BANG_Base/bang_search.cu:138-362)
plus the query / ground-truth files of the harness (BANG_Base/test_driver.cpp:238-272,
353-373).  Layouts, all little-endian:

``<p>_pq_compressed.bin``  {i32 N, i32 m, u8[N][m]}                       (bang_search.cu:218-234)
``<p>_pq_pivots.bin``      {u32 4, u32 1, u64 off_pivots, u64 off_centroid, u64 off_chunkoffs,
                            u64 file_size}; at every offset an 8-byte {i32 rows, i32 cols}
                            header then data: pivots f32[256][D], centroid f32[D],
                            chunk offsets u32[m+1]                           (bang_search.cu:246-296)
``<p>_disk.bin``           N entries of entryLen = D*sizeof(T)+4+4R bytes:
                            [T vec[D]][u32 degree][u32 nbr[R]]               (bang_search.cu:335-340)
``<p>_disk_metadata.bin``  packed 32 B {u64 medoid, u64 entryLen, i32 dtype, u32 D, u32 R,
                            u32 N}                                           (bang_search.cuh:42-50)
query ``.bin``             {i32 n, i32 D, T[n][D]}                          (test_driver.cpp:360-373)
ground truth ``.bin``      {i32 n, i32 K, u32[n][K], f32[n][K]}             (test_driver.cpp:238-272)
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

PQ_PIVOTS_SUFFIX = "_pq_pivots.bin"
PQ_COMPRESSED_SUFFIX = "_pq_compressed.bin"
GRAPH_SUFFIX = "_disk.bin"
GRAPH_META_SUFFIX = "_disk_metadata.bin"

# dtype codes as written by bang_preprocess.py:12-13 (0=int8, 1=uint8, 2=float).
# The engine never reads the field; it is kept for fidelity.
PREPROCESS_DTYPE_CODE = {"int8": 0, "uint8": 1, "float": 2}
NP_DTYPE = {"uint8": np.uint8, "int8": np.int8, "float": np.float32}
PIVOTS_METADATA_SIZE = 4096  # DiskANN pads the offset table to one sector


def entry_len(D: int, R: int, dtype: str) -> int:
    return D * np.dtype(NP_DTYPE[dtype]).itemsize + 4 + 4 * R


@dataclass
class Index:
    """In-memory image of the four index files."""
    dtype: str                # "uint8" | "int8" | "float"
    N: int
    D: int
    R: int
    m: int
    medoid: int
    graph: np.ndarray         # uint8 [N, entryLen]
    codes: np.ndarray         # uint8 [N, m]
    pivots: np.ndarray        # float32 [256, D]  (file order, NOT transposed)
    centroid: np.ndarray      # float32 [D]
    chunk_off: np.ndarray     # uint32 [m+1]

    @property
    def entry_len(self) -> int:
        return entry_len(self.D, self.R, self.dtype)

    def vectors(self) -> np.ndarray:
        isz = np.dtype(NP_DTYPE[self.dtype]).itemsize
        return self.graph[:, : self.D * isz].copy().view(NP_DTYPE[self.dtype]).reshape(self.N, self.D)

    def degrees(self) -> np.ndarray:
        isz = np.dtype(NP_DTYPE[self.dtype]).itemsize
        o = self.D * isz
        return self.graph[:, o:o + 4].copy().view(np.uint32).reshape(self.N)

    def adjacency(self) -> np.ndarray:
        isz = np.dtype(NP_DTYPE[self.dtype]).itemsize
        o = self.D * isz + 4
        return self.graph[:, o:o + 4 * self.R].copy().view(np.uint32).reshape(self.N, self.R)


def pack_graph(vectors: np.ndarray, degrees: np.ndarray, adjacency: np.ndarray) -> np.ndarray:
    """[T vec[D]][u32 deg][u32 nbr[R]] per node -> uint8 [N, entryLen]."""
    N, D = vectors.shape
    R = adjacency.shape[1]
    isz = vectors.dtype.itemsize
    out = np.zeros((N, D * isz + 4 + 4 * R), dtype=np.uint8)
    out[:, : D * isz] = np.ascontiguousarray(vectors).view(np.uint8).reshape(N, D * isz)
    out[:, D * isz: D * isz + 4] = np.ascontiguousarray(degrees.astype("<u4")).view(np.uint8).reshape(N, 4)
    out[:, D * isz + 4:] = np.ascontiguousarray(adjacency.astype("<u4")).view(np.uint8).reshape(N, 4 * R)
    return out


def write_pq_compressed(path: str, codes: np.ndarray) -> None:
    N, m = codes.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", N, m))
        f.write(np.ascontiguousarray(codes, dtype=np.uint8).tobytes())


def read_pq_compressed(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        N, m = struct.unpack("<ii", f.read(8))
        return np.frombuffer(f.read(N * m), dtype=np.uint8).reshape(N, m).copy()


def write_pq_pivots(path: str, pivots: np.ndarray, centroid: np.ndarray, chunk_off: np.ndarray) -> None:
    """DiskANN-style sectioned file; bang_load reads each section at offset+8."""
    D = pivots.shape[1]
    m = chunk_off.shape[0] - 1
    assert pivots.shape == (256, D) and centroid.shape == (D,)
    off_piv = PIVOTS_METADATA_SIZE
    off_cen = off_piv + 8 + 256 * D * 4
    off_chk = off_cen + 8 + D * 4
    fsize = off_chk + 8 + (m + 1) * 4
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 4, 1))
        f.write(struct.pack("<QQQQ", off_piv, off_cen, off_chk, fsize))
        f.write(b"\0" * (PIVOTS_METADATA_SIZE - f.tell()))
        f.write(struct.pack("<ii", 256, D))
        f.write(np.ascontiguousarray(pivots, dtype="<f4").tobytes())
        f.write(struct.pack("<ii", D, 1))
        f.write(np.ascontiguousarray(centroid, dtype="<f4").tobytes())
        f.write(struct.pack("<ii", m + 1, 1))
        f.write(np.ascontiguousarray(chunk_off, dtype="<u4").tobytes())
        assert f.tell() == fsize


def read_pq_pivots(path: str, D: int, m: int):
    with open(path, "rb") as f:
        nsec, _ = struct.unpack("<II", f.read(8))
        if nsec != 4:
            raise ValueError("PQ pivots file does not contain the required number of sections")
        off_piv, off_cen, off_chk, _fsize = struct.unpack("<QQQQ", f.read(32))
        f.seek(off_piv + 8)
        pivots = np.frombuffer(f.read(256 * D * 4), dtype="<f4").reshape(256, D).copy()
        f.seek(off_cen + 8)
        centroid = np.frombuffer(f.read(D * 4), dtype="<f4").copy()
        f.seek(off_chk + 8)
        chunk_off = np.frombuffer(f.read((m + 1) * 4), dtype="<u4").copy()
    return pivots, centroid, chunk_off


def write_graph_metadata(path: str, medoid: int, entry_len_: int, dtype: str, D: int, R: int, N: int) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<QQiIII", medoid, entry_len_, PREPROCESS_DTYPE_CODE[dtype], D, R, N))


def read_graph_metadata(path: str) -> dict:
    with open(path, "rb") as f:
        medoid, elen, dt, D, R, N = struct.unpack("<QQiIII", f.read(32))
    return dict(medoid=medoid, entry_len=elen, dtype_code=dt, D=D, R=R, N=N)


def write_index(prefix: str, ix: Index) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    write_pq_pivots(prefix + PQ_PIVOTS_SUFFIX, ix.pivots, ix.centroid, ix.chunk_off)
    write_pq_compressed(prefix + PQ_COMPRESSED_SUFFIX, ix.codes)
    with open(prefix + GRAPH_SUFFIX, "wb") as f:
        f.write(np.ascontiguousarray(ix.graph).tobytes())
    write_graph_metadata(prefix + GRAPH_META_SUFFIX, ix.medoid, ix.entry_len, ix.dtype, ix.D, ix.R, ix.N)


def read_index(prefix: str, dtype: str, mmap_graph: bool = False) -> Index:
    """mmap_graph=True maps `_disk.bin` read-only instead of reading it (the ranks of a multi-GPU job share one copy)."""
    md = read_graph_metadata(prefix + GRAPH_META_SUFFIX)
    codes = read_pq_compressed(prefix + PQ_COMPRESSED_SUFFIX)
    N, m = codes.shape
    pivots, centroid, chunk_off = read_pq_pivots(prefix + PQ_PIVOTS_SUFFIX, md["D"], m)
    if mmap_graph:
        graph = np.memmap(prefix + GRAPH_SUFFIX, dtype=np.uint8, mode="r").reshape(md["N"], md["entry_len"])
    else:
        graph = np.fromfile(prefix + GRAPH_SUFFIX, dtype=np.uint8).reshape(md["N"], md["entry_len"])
    return Index(dtype=dtype, N=md["N"], D=md["D"], R=md["R"], m=m, medoid=md["medoid"], graph=graph,
                 codes=codes, pivots=pivots, centroid=centroid, chunk_off=chunk_off)


def write_bin(path: str, data: np.ndarray) -> None:
    """{i32 n, i32 D, T[n][D]} -- query files, base files."""
    n, d = data.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n, d))
        f.write(np.ascontiguousarray(data).tobytes())


def read_bin(path: str, dtype: str) -> np.ndarray:
    with open(path, "rb") as f:
        n, d = struct.unpack("<ii", f.read(8))
        return np.frombuffer(f.read(), dtype=NP_DTYPE[dtype], count=n * d).reshape(n, d).copy()


def write_truthset(path: str, ids: np.ndarray, dists: np.ndarray) -> None:
    n, K = ids.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n, K))
        f.write(np.ascontiguousarray(ids, dtype="<u4").tobytes())
        f.write(np.ascontiguousarray(dists, dtype="<f4").tobytes())


def read_truthset(path: str):
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        n, K = struct.unpack("<ii", f.read(8))
        if size != 2 * n * K * 4 + 8:  # test_driver.cpp:254-266
            raise ValueError(f"truthset size mismatch: {size} vs {2 * n * K * 4 + 8}")
        ids = np.frombuffer(f.read(n * K * 4), dtype="<u4").reshape(n, K).copy()
        dists = np.frombuffer(f.read(n * K * 4), dtype="<f4").reshape(n, K).copy()
    return ids, dists


# ---------------------------------------------------------------------------------------------
# DiskANN `_disk.index`  ->  `_disk.bin` + `_disk_metadata.bin`   (reference: bang_preprocess.py)
# ---------------------------------------------------------------------------------------------
SECTOR_LEN = 4096  # bang_preprocess.py:21


def write_diskann_index(path: str, vectors: np.ndarray, degrees: np.ndarray, adjacency: np.ndarray, medoid: int,
                        pad_garbage: bool = False) -> None:
    """Write a DiskANN-style sector-padded `_disk.index` (test input for the converter).
    Sector 0 = header: {u32,u32 (skipped), u64 npts, u64 ndims, u64 medoid, u64 max_node_len, u64 nnodes_per_sector,
    3 x u64 (skipped), u64 file_size}; then ``nnodes_per_sector`` node records per 4096-byte sector, each
    [T vec[D]][u32 degree][u32 nbr[R]] with the adjacency in ARBITRARY (unsorted) order (bang_preprocess.py:28-109)."""
    N, D = vectors.shape
    R = adjacency.shape[1]
    node_len = D * vectors.dtype.itemsize + 4 + 4 * R
    per_sector = SECTOR_LEN // node_len
    n_sectors = (N + per_sector - 1) // per_sector
    fsize = (n_sectors + 1) * SECTOR_LEN
    rng = np.random.default_rng(7)
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 9, 1))
        f.write(struct.pack("<QQQQQ", N, D, medoid, node_len, per_sector))
        f.write(struct.pack("<QQQ", 0, 0, 0))
        f.write(struct.pack("<Q", fsize))
        f.write(b"\0" * (SECTOR_LEN - f.tell()))
        for s in range(n_sectors):
            buf = bytearray(SECTOR_LEN)
            for j in range(per_sector):
                i = s * per_sector + j
                if i >= N:
                    break
                deg = int(degrees[i])
                nb = adjacency[i, :deg].astype("<u4").copy()
                rng.shuffle(nb)                     # DiskANN does not sort; the converter must
                rec = vectors[i].tobytes() + struct.pack("<I", deg) + nb.tobytes()
                if pad_garbage:
                    rec += rng.integers(1, 2**32, R - deg, dtype=np.uint64).astype("<u4").tobytes()
                rec += b"\0" * (node_len - len(rec))
                buf[j * node_len:(j + 1) * node_len] = rec
            f.write(bytes(buf))


def convert_diskann_index(index_path: str, out_bin_path: str, D: int, dtype: str, R: int) -> dict:
    """Compact a DiskANN `_disk.index` into `<x>_disk.bin` + `<x>_disk_metadata.bin`, sorting every adjacency list
    ascending -- same outputs as the reference's bang_preprocess.py (header parse order :28-64, sector walk :75-80,
    per-node copy + sort :81-109, metadata order :42,47,49-51,116), vectorised with numpy."""
    isz = np.dtype(NP_DTYPE[dtype]).itemsize
    meta_path = out_bin_path[:-4] + "_metadata" + out_bin_path[-4:]
    with open(index_path, "rb") as f:
        f.read(8)
        npts, ndims, medoid, max_node_len, per_sector = struct.unpack("<QQQQQ", f.read(40))
        f.read(24)
        (fsize,) = struct.unpack("<Q", f.read(8))
        n_sectors = fsize // SECTOR_LEN - 1
        node_len = D * isz + 4 + 4 * R
        out = np.zeros((npts, node_len), dtype=np.uint8)
        done = 0
        for s in range(n_sectors):
            f.seek((s + 1) * SECTOR_LEN)
            sec = np.frombuffer(f.read(SECTOR_LEN), dtype=np.uint8)
            take = min(per_sector, npts - done)
            if take <= 0:
                break
            recs = sec[: take * max_node_len].reshape(take, max_node_len)[:, :node_len]
            out[done:done + take] = recs
            done += take
    deg = out[:, D * isz:D * isz + 4].copy().view("<u4").reshape(-1)
    if (deg > R).any() or (deg == 0).any():
        raise ValueError("bad degree in index (bang_preprocess.py:91-94)")
    adj = out[:, D * isz + 4:].copy().view("<u4").reshape(npts, R)
    col = np.arange(R)[None, :]
    big = np.where(col < deg[:, None], adj, np.uint32(0xFFFFFFFF))
    order = np.argsort(big, axis=1, kind="stable")
    srt = np.take_along_axis(adj, order, axis=1)             # valid ids ascending first, padding after
    out[:, D * isz + 4:] = np.ascontiguousarray(srt.astype("<u4")).view(np.uint8).reshape(npts, 4 * R)
    with open(out_bin_path, "wb") as w:
        w.write(out.tobytes())
    with open(meta_path, "wb") as w:
        w.write(struct.pack("<QQiIII", medoid, max_node_len, PREPROCESS_DTYPE_CODE[dtype], D, R, done))
    return dict(npts=npts, ndims=ndims, medoid=medoid, max_node_len=max_node_len, nodes=done)
