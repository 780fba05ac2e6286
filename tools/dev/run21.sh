#!/bin/bash
set -u
# phase probe build (lib_ab/libbang_pp.so): per-iteration phase times, device graph, full and shard-sized batches
export BANG_AMD_LIB=$PWD/bang-billion-scale-ann_amd/lib_ab/libbang_pp.so
for g in ${GRAPHS:-device}; do for q in ${QS:-10000 1250}; do
  BANG_SEARCH_PROF=1 timeout 300 python bench.py --graph $g --queries $q --no-legs --no-cpu-baseline --steps 3 --warmup 1 --L 70 > gpurun_out/b21_${g}_$q.json 2> gpurun_out/b21_${g}_$q.err
  echo "== $g $q"
  grep "\[search\]" gpurun_out/b21_${g}_$q.err | tail -2 | cut -c1-400
done; done
