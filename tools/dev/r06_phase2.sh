#!/bin/bash
# phase times of two diagnostic builds side by side (LIBS: directories under the package holding a -DBANG_SEARCH_PHASE_PROF library)
cd "$(dirname "$0")/../.."
O=gpurun_out/${TAG:-r06_phase2}; mkdir -p $O
PKG=$PWD/bang-billion-scale-ann_amd
for lib in $LIBS; do
  BANG_AMD_LIB=$PKG/$lib/libbang.so BANG_NO_BUILD=1 BANG_SEARCH_PROF=1 timeout 1200 python tools/shard_sweep.py --queries ${QUERIES:-10000,2500,1250} --variants default --steps 3 --out $O/$lib.md > $O/$lib.log 2> $O/$lib.err
  echo "== $lib"; grep '^| [0-9]' $O/$lib.md; grep "phases of an iteration" $O/$lib.err | awk 'NR%4==0'
done
