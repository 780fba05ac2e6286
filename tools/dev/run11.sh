#!/bin/bash
set -u
# diagnostic build (phase probe) -- numbers are perturbed by the stamps
for g in device host; do
  BANG_SEARCH_PROF=1 timeout 300 python bench.py --graph $g --no-legs --no-cpu-baseline --steps 3 --warmup 1 --L 70 > gpurun_out/b11_$g.json 2> gpurun_out/b11_$g.err
  grep "\[search\]" gpurun_out/b11_$g.err | tail -2
  python - <<PY
import json
d=json.loads(open('gpurun_out/b11_$g.json').read().strip().splitlines()[-1])
print("$g:", d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
PY
done
BANG_SEARCH_PROF=1 timeout 300 python bench.py --workload deep100m_shape --no-legs --steps 2 --warmup 1 > gpurun_out/b11_deep.json 2> gpurun_out/b11_deep.err
grep "\[search\]" gpurun_out/b11_deep.err | tail -1
