#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 300 tools/dev/rand_sector_bench > gpurun_out/rand_sector.jsonl 2>&1
cat gpurun_out/rand_sector.jsonl
for w in 8 12; do
  BANG_SEARCH_MAX_WAVES=$w timeout 600 python bench.py --graph device --no-legs --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/b4_w$w.json 2> gpurun_out/b4_w$w.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/b4_w$w.json').read().strip().splitlines()[-1])
print("waves $w:", d['value'], d['ms_per_step'], d['roofline'] and d['roofline']['avg_launch_us'])
PY
done
