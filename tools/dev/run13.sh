#!/bin/bash
set -u
# diagnostic build (phase probe): where an iteration spends its time when the GPU is NOT full (one rank's shard of the batch)
for g in device host; do for q in 10000 2500 1250; do
  BANG_SEARCH_PROF=1 timeout 300 python bench.py --graph $g --queries $q --no-legs --no-cpu-baseline --steps 3 --warmup 1 --L 70 > gpurun_out/b13_${g}_$q.json 2> gpurun_out/b13_${g}_$q.err
  echo "== $g $q"
  grep "\[search\]" gpurun_out/b13_${g}_$q.err | tail -2 | cut -c1-400
  python - <<PY
import json
d=json.loads(open('gpurun_out/b13_${g}_$q.json').read().strip().splitlines()[-1])
print("$g $q:", d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], {k:v for k,v in d['config'].items() if 'iter' in k})
PY
done; done
