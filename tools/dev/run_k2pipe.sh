#!/bin/bash
# K2 alone: the row reduce as it is vs the two-stage LDS pipeline (lib_k2pipe: make OUT=lib_k2pipe EXTRA_CXXFLAGS=-DBANG_K2_REDUCE_PIPE=1)
cd "$(dirname "$0")/../.."
fmt='
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("m",d["m"],"stride",d["code_stride"],"us",d["avg_launch_us"],"rows/s",d["rows_per_s"],"frac",d["frac"])'
for rep in 1 2; do
echo "== base"; python tools/k2_alone.py --big 2>/dev/null | python -c "$fmt"
echo "== pipe"; BANG_AMD_LIB=$PWD/bang-billion-scale-ann_amd/lib_k2pipe/libbang.so python tools/k2_alone.py --big 2>/dev/null | python -c "$fmt"
done
BANG_AMD_LIB=$PWD/bang-billion-scale-ann_amd/lib_k2pipe/libbang.so python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
