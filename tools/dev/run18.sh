#!/bin/bash
set -u
( time timeout 500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py tests/test_golden.py -x -q ) > gpurun_out/t18.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/t18.log
for g in device host; do
  timeout 300 python bench.py --graph $g --no-legs --no-cpu-baseline --steps 10 --warmup 3 --L 70 > gpurun_out/b18_$g.json 2> gpurun_out/b18_$g.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/b18_$g.json').read().strip().splitlines()[-1])
print("$g:", d['value'], d['ms_per_step'], d['config']['parity_vs_oracle_first_64'], d['roofline']['avg_launch_us'])
PY
done
for w in deep100m_shape sift1b_shape; do
  timeout 400 python bench.py --workload $w --no-legs --steps 5 --warmup 2 > gpurun_out/b18_$w.json 2> gpurun_out/b18_$w.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/b18_$w.json').read().strip().splitlines()[-1])
print("$w:", d['value'], d['ms_per_step'], d['config'].get('result_properties_ok'), d['roofline']['avg_launch_us'])
PY
done
