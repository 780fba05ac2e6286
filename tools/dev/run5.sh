#!/bin/bash
set -u
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_engine.py -x -q -k "search_kernel or host_loop" ) > gpurun_out/t5.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/t5.log
for s in 1 0; do
  BANG_SEARCH=$s timeout 600 python bench.py --graph host --no-legs --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/b5_host_s$s.json 2> gpurun_out/b5_host_s$s.err
  echo "search=$s rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b5_host_s$s.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['config']['iterations'], d['config']['parity_vs_oracle_first_64'], d['roofline'] and d['roofline']['avg_launch_us'], d['config']['step_ms'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/b5_host_s$s.err').read()[-1500:])
PY
done
for s in 1 0; do
  BANG_SEARCH=$s timeout 900 python bench.py --workload sift1b_shape --no-legs --steps 5 --warmup 2 > gpurun_out/b5_1b_s$s.json 2> gpurun_out/b5_1b_s$s.err
  echo "sift1b search=$s rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b5_1b_s$s.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['config']['iterations'], d['config'].get('result_properties_ok'), d['roofline'] and d['roofline']['avg_launch_us'], d['roofline'] and d['roofline'].get('pcie_h2d'))
except Exception as e:
    print('ERR', e); print(open('gpurun_out/b5_1b_s$s.err').read()[-1500:])
PY
done
