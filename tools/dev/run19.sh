#!/bin/bash
set -u
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/t19.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/t19.log
for g in device host; do for q in 10000 5000 2500 1250; do
  timeout 300 python bench.py --graph $g --queries $q --L 70 --no-legs --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/b19_${g}_$q.json 2> gpurun_out/b19_${g}_$q.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b19_${g}_$q.json').read().strip().splitlines()[-1])
    print("$g Q=$q:", d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
except Exception as e:
    print("$g $q error", e)
PY
done; done
for w in deep100m_shape sift1b_shape; do
  timeout 600 python bench.py --workload $w --no-legs --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/b19_$w.json 2> gpurun_out/b19_$w.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b19_$w.json').read().strip().splitlines()[-1])
    print("$w:", d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
except Exception as e:
    print("$w error", e)
PY
done
