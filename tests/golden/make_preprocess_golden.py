"""Pins our DiskANN-index converter (bang_amd.formats.convert_diskann_index) to the REFERENCE's own converter.

Run in the build container only (it executes /root/reference/BANG_Base/bang_preprocess.py, which never travels):

    python tests/golden/make_preprocess_golden.py

For every dtype it writes a tiny sector-padded `_disk.index` (ragged degrees, shuffled adjacency lists, garbage in the
unused tail of every list, several sectors with a partly filled last one), runs the reference script on it and commits
the script's two output files as fixtures next to the input:

    tests/golden/pre_<dtype>_disk.index           input  (made by our writer, formats.write_diskann_index)
    tests/golden/pre_<dtype>_disk.bin             output of the reference's bang_preprocess.py
    tests/golden/pre_<dtype>_disk_metadata.bin    output of the reference's bang_preprocess.py

tests/test_formats.py::test_converter_matches_reference_preprocess compares our converter with them byte for byte."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))

from bang_amd import formats  # noqa: E402

REF = "/root/reference/BANG_Base/bang_preprocess.py"
CASES = {   # dtype: (N, D, R) -- node records of 76 / 52 / 84 bytes: 53 / 78 / 48 per 4096-byte sector
    "uint8": (130, 24, 12),
    "int8": (170, 16, 8),
    "float": (101, 12, 8),
}


def make_case(dtype, N, D, R, seed):
    rng = np.random.default_rng(seed)
    if dtype == "float":
        vec = rng.normal(size=(N, D)).astype(np.float32)
    elif dtype == "uint8":
        vec = rng.integers(0, 256, (N, D), dtype=np.uint8)
    else:
        vec = rng.integers(-128, 128, (N, D), dtype=np.int8)
    deg = rng.integers(1, R + 1, N).astype(np.uint32)          # ragged: 1..R (0 is rejected by the reference, :91-94)
    deg[0], deg[1] = R, 1
    adj = np.zeros((N, R), np.uint32)
    for i in range(N):
        adj[i, : deg[i]] = np.sort(rng.choice(N, deg[i], replace=False)).astype(np.uint32)
    return vec, deg, adj


def main():
    if not os.path.exists(REF):
        raise SystemExit(f"{REF} not found: this generator only runs where the reference checkout is present")
    for n, (dtype, (N, D, R)) in enumerate(CASES.items()):
        vec, deg, adj = make_case(dtype, N, D, R, 4100 + n)
        stem = os.path.join(HERE, f"pre_{dtype}")
        formats.write_diskann_index(stem + "_disk.index", vec, deg, adj, medoid=N // 3, pad_garbage=True)
        r = subprocess.run([sys.executable, REF, stem + "_disk.index", stem + "_disk.bin", str(D),
                            str(formats.PREPROCESS_DTYPE_CODE[dtype]), str(R)], capture_output=True, text=True)
        if r.returncode != 0 or f"Total # of Nodes Discovered = {N}" not in r.stdout:
            raise SystemExit(f"reference preprocess failed for {dtype}:\n{r.stdout[-800:]}\n{r.stderr[-800:]}")
        print(dtype, {s: os.path.getsize(stem + s) for s in ("_disk.index", "_disk.bin", "_disk_metadata.bin")})


if __name__ == "__main__":
    main()
