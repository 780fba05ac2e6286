#!/bin/bash
set -u
mkdir -p gpurun_out
( time timeout 400 python -m pytest tests/test_gpu_engine.py -x -q -k "search_kernel_host" ) > gpurun_out/t8.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/t8.log
for c in 2 1; do
  BANG_SEARCH_CTX=$c BANG_WALK_PROF=1 timeout 300 python bench.py --graph host --no-legs --no-cpu-baseline --steps 10 --warmup 3 --L 70 > gpurun_out/b8_c$c.json 2> gpurun_out/b8_c$c.err
  grep "\[walk\]" gpurun_out/b8_c$c.err | tail -3
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b8_c$c.json').read().strip().splitlines()[-1])
    print("ctx $c:", d['value'], d['ms_per_step'], d['config']['parity_vs_oracle_first_64'], d['roofline']['avg_launch_us'], d['config']['step_ms'], d['config']['queries_per_workgroup'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/b8_c$c.err').read()[-1500:])
PY
done
for c in 2 1; do
  BANG_SEARCH_CTX=$c timeout 400 python bench.py --workload sift1b_shape --no-legs --steps 5 --warmup 2 > gpurun_out/b8_1b_c$c.json 2> gpurun_out/b8_1b_c$c.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b8_1b_c$c.json').read().strip().splitlines()[-1])
    print("sift1b ctx $c:", d['value'], d['ms_per_step'], d['config'].get('result_properties_ok'), d['roofline']['avg_launch_us'], d['roofline'].get('pcie_h2d',{}).get('achieved_GBps'), d['config']['queries_per_workgroup'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/b8_1b_c$c.err').read()[-1500:])
PY
done
