#!/bin/bash
set -u
( time timeout 400 python -m pytest tests/test_gpu_engine.py tests/test_golden.py -x -q ) > gpurun_out/t14.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/t14.log
for g in device host; do
  timeout 300 python bench.py --graph $g --no-legs --no-cpu-baseline --steps 10 --warmup 3 --L 70 > gpurun_out/b14_$g.json 2> gpurun_out/b14_$g.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/b14_$g.json').read().strip().splitlines()[-1])
print("$g:", d['value'], d['ms_per_step'], d['config']['parity_vs_oracle_first_64'], d['roofline']['avg_launch_us'])
PY
done
