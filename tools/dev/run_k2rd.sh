#!/bin/bash
# K2 alone: code rows of 2 (default) / 3 / 4 neighbour rows in flight per wave (lib_k2rd3, lib_k2rd4: make OUT=... EXTRA_CXXFLAGS=-DBANG_K2_RD=3)
cd "$(dirname "$0")/../.."
fmt='
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("m",d["m"],"stride",d["code_stride"],"us",d["avg_launch_us"],"min",d["min_launch_us"],"rows/s",d["rows_per_s"],"frac",d["frac"])'
for rep in 1 2; do
for v in "" lib_k2rd3 lib_k2rd4; do
  echo "== ${v:-default (2)}"
  if [ -z "$v" ]; then python tools/k2_alone.py --big 2>/dev/null | python -c "$fmt"; else BANG_AMD_LIB=$PWD/bang-billion-scale-ann_amd/$v/libbang.so python tools/k2_alone.py --big 2>/dev/null | python -c "$fmt"; fi
done
done
