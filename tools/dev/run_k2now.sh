#!/bin/bash
# K2 alone at HEAD, twice, + the kernel parity tests
cd "$(dirname "$0")/../.."
fmt='
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("m",d["m"],"stride",d["code_stride"],"us",d["avg_launch_us"],"min",d["min_launch_us"],"rows/s",d["rows_per_s"],"frac",d["frac"])'
for rep in 1 2; do python tools/k2_alone.py --big 2>/dev/null | python -c "$fmt"; done
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
