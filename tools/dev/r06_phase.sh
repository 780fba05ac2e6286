#!/bin/bash
# phase times of an iteration (diagnostic build lib_prof: -DBANG_SEARCH_PHASE_PROF -DBANG_DEV_ONLY_218), SIFT1B shape: rows pulled (N = 1e9) and all rows in HBM (N = 5e8)
cd "$(dirname "$0")/../.."
O=gpurun_out/${TAG:-r06_phase}; mkdir -p $O
PKG=$PWD/bang-billion-scale-ann_amd
export BANG_AMD_LIB=$PKG/lib_prof/libbang.so BANG_NO_BUILD=1 BANG_SEARCH_PROF=1
timeout 1200 python tools/shard_sweep.py --queries 10000,2500,1250 --variants default --steps 3 --out $O/pulled.md > $O/pulled.log 2> $O/pulled.err
echo "== rows pulled, N = 1e9"; grep "phases of an iteration" $O/pulled.err | awk 'NR%4==0'
timeout 1200 python tools/shard_sweep.py --shape-n 500000000 --queries 10000,2500,1250 --variants rows_frac=1.0 --steps 3 --out $O/hbm.md > $O/hbm.log 2> $O/hbm.err
echo "== all rows in HBM, N = 5e8"; grep "phases of an iteration" $O/hbm.err | awk 'NR%4==0'
