#!/bin/bash
# per-iteration phase times of the search kernel (diagnostic build: make OUT=lib_prof EXTRA_CXXFLAGS=-DBANG_SEARCH_PHASE_PROF), full batch and a
# 1 250-query shard, SIFT1B-shape + DEEP100M-shape
cd /root/repo
mkdir -p gpurun_out/phase
ALT=$PWD/bang-billion-scale-ann_amd/lib_prof/libbang.so
for spec in "sift1b_shape 10000" "sift1b_shape 1250" "deep100m_shape 10000" "deep100m_shape 1250" "sift1m 10000"; do
  set -- $spec
  BANG_AMD_LIB=$ALT BANG_SEARCH_PROF=1 timeout 900 python3 bench.py --workload $1 --queries $2 --no-legs --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1 \
      > gpurun_out/phase/$1_$2.json 2> gpurun_out/phase/$1_$2.err
  echo "== $1 Q=$2: $(python3 -c "import json,sys; d=json.loads([l for l in open('gpurun_out/phase/$1_$2.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'])")"
  grep "phases of an iteration" gpurun_out/phase/$1_$2.err | tail -2
done
