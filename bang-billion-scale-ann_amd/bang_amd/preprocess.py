"""Command-line counterpart of the reference's BANG_Base/bang_preprocess.py (same five positional arguments):

    python -m bang_amd.preprocess <DiskANN _disk.index> <output _disk.bin> <dimension> <datatype: 0 int8, 1 uint8, 2 float> <R>

Writes <output> (contiguous [vec][degree][sorted neighbours] entries) and <output minus .bin>_metadata.bin (32-byte packed
metadata) -- the two files bang_load needs next to DiskANN's _pq_pivots.bin / _pq_compressed.bin."""
import sys

from .formats import convert_diskann_index

CODE_TO_DTYPE = {0: "int8", 1: "uint8", 2: "float"}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 5:
        print("Usage: python -m bang_amd.preprocess <path to DiskANN graph index file (.index)> <path to store the o/p file "
              "for BANG search (.bin)> <dataset dimension> <dataset datatype: 0 -> int8, 1 -> uint8, 2 -> float> "
              "<degree (i.e. R) of the DiskANN graph index>")
        return 2
    src, dst, dim, code, R = argv[0], argv[1], int(argv[2]), int(argv[3]), int(argv[4])
    info = convert_diskann_index(src, dst, dim, CODE_TO_DTYPE[code], R)
    print("Number of Nodes: ", info["npts"])
    print("Dataset Dimensions: ", info["ndims"])
    print("Medoid: ", info["medoid"])
    print("Each node entry length (bytes):", info["max_node_len"])
    print("Total # of Nodes Discovered =", info["nodes"])
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
