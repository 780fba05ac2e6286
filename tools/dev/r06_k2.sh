#!/bin/bash
# K2 alone (tools/k2_alone.py --big: 40 M evaluations per launch, 4 GB code table, five layouts) per library directory, interleaved twice.
cd "$(dirname "$0")/../.."
O=gpurun_out/${TAG:-r06_k2}; mkdir -p $O
PKG=$PWD/bang-billion-scale-ann_amd
for pass in 1 2; do for lib in ${LIBS:-lib}; do
  [ -f $PKG/$lib/libbang.so ] || { echo "no $lib"; continue; }
  BANG_AMD_LIB=$PKG/$lib/libbang.so BANG_NO_BUILD=1 timeout 600 python tools/k2_alone.py --big > $O/k2_${lib}_$pass.jsonl 2> $O/k2_${lib}_$pass.err
  echo "== $lib pass $pass"
  python - $O/k2_${lib}_$pass.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"m={d['m']} stride={d['code_stride']}: {d['avg_launch_us']} us (min {d['min_launch_us']}) {d['rows_per_s']} G rows/s frac {d['frac']:.4f}")
P
done; done
