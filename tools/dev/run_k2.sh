#!/bin/bash
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
echo "== coop"; python tools/k2_alone.py --big 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('m',d['m'],'stride',d['code_stride'],'us',d['avg_launch_us'],'rows/s',d['rows_per_s'],'frac',d['frac'])"
echo "== per-lane"; BANG_AMD_LIB=$PWD/bang-billion-scale-ann_amd/lib_k2lane/libbang.so python tools/k2_alone.py --big 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('m',d['m'],'stride',d['code_stride'],'us',d['avg_launch_us'],'rows/s',d['rows_per_s'],'frac',d['frac'])"
