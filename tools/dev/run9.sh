#!/bin/bash
set -u
for c in 1 2; do
  BANG_SEARCH_PROF=1 BANG_SEARCH_CTX=$c timeout 300 python bench.py --graph host --no-legs --no-cpu-baseline --steps 4 --warmup 2 --L 70 > gpurun_out/b9_c$c.json 2> gpurun_out/b9_c$c.err
  grep "\[search\]" gpurun_out/b9_c$c.err | tail -2
  python - <<PY
import json
d=json.loads(open('gpurun_out/b9_c$c.json').read().strip().splitlines()[-1])
print("ctx $c:", d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
PY
done
