#!/bin/bash
set -u
mkdir -p gpurun_out/profiles_out
df -h /dev/shm | tail -1
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/t12.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/t12.log
# what a rank's shard of the 10K batch takes on a GPU of its own (strong-scaling projection), both graph placements
for g in device host; do for q in 10000 5000 2500 1250; do
  timeout 300 python bench.py --graph $g --queries $q --L 70 --no-legs --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/b12_${g}_$q.json 2> gpurun_out/b12_${g}_$q.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b12_${g}_$q.json').read().strip().splitlines()[-1])
    print("$g Q=$q:", d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
except Exception as e:
    print('ERR $g $q', e)
PY
done; done
