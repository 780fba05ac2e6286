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


def read_index(prefix: str, dtype: str) -> Index:
    md = read_graph_metadata(prefix + GRAPH_META_SUFFIX)
    codes = read_pq_compressed(prefix + PQ_COMPRESSED_SUFFIX)
    N, m = codes.shape
    pivots, centroid, chunk_off = read_pq_pivots(prefix + PQ_PIVOTS_SUFFIX, md["D"], m)
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
