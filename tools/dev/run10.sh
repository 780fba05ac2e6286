#!/bin/bash
set -u
mkdir -p gpurun_out
( time timeout 300 python -m pytest tests/test_gpu_engine.py -x -q -k "search_kernel_host" ) > gpurun_out/t10.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/t10.log
( BANG_SEARCH_GS=4 timeout 300 python -m pytest tests/test_gpu_engine.py -x -q -k "search_kernel_host" ) > gpurun_out/t10b.log 2>&1
echo "pytest gs4 rc=$?"; tail -2 gpurun_out/t10b.log
for g in 16 8 4; do
  BANG_SEARCH_PROF=1 BANG_SEARCH_GS=$g BANG_WALK_PROF=1 timeout 300 python bench.py --graph host --no-legs --no-cpu-baseline --steps 6 --warmup 2 --L 70 > gpurun_out/b10_g$g.json 2> gpurun_out/b10_g$g.err
  grep "\[search\]" gpurun_out/b10_g$g.err | tail -1
  grep "\[walk\]" gpurun_out/b10_g$g.err | tail -2
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b10_g$g.json').read().strip().splitlines()[-1])
    print("gs $g:", d['value'], d['ms_per_step'], d['config']['parity_vs_oracle_first_64'], d['roofline']['avg_launch_us'], d['config']['step_ms'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/b10_g$g.err').read()[-1500:])
PY
done
for g in 16 8; do
  BANG_SEARCH_GS=$g timeout 400 python bench.py --workload sift1b_shape --no-legs --steps 5 --warmup 2 > gpurun_out/b10_1b_g$g.json 2> gpurun_out/b10_1b_g$g.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b10_1b_g$g.json').read().strip().splitlines()[-1])
    print("sift1b gs $g:", d['value'], d['ms_per_step'], d['config'].get('result_properties_ok'), d['roofline']['avg_launch_us'], d['roofline'].get('pcie_h2d',{}).get('achieved_GBps'))
except Exception as e:
    print('ERR', e); print(open('gpurun_out/b10_1b_g$g.err').read()[-1500:])
PY
done
