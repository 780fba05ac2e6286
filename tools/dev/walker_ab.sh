#!/bin/bash
# walker form (north-star data flow) on the SIFT1B-shape index with a resident graph image: memcpy vs 512-bit NT stores, thread counts
cd "$(dirname "$0")/../.."
for rep in 1 2; do for v in "BANG_WALK_NT=0" "BANG_WALK_NT=1" "BANG_THREADS=8" "BANG_THREADS=14"; do
  env $v python bench.py --workload sift1b_shape --pull 0 --resident-graph --no-legs --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v', j['value'], j['ms_per_step'], 'min/max', j['config']['step_ms_min'], j['config']['step_ms_max'], 'BAR GB/s', (r.get('pcie_h2d') or {}).get('achieved_GBps'), 'threads', j['config']['walker_threads'])"
done; done
