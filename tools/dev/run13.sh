#!/bin/bash
set -u
mkdir -p gpurun_out/profiles_out
( time timeout 200 python -m pytest tests/test_gpu_multirank.py -x -q ) > gpurun_out/t13.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/t13.log
for g in device host; do
( time timeout 900 python bench.py --workload sift10m --graph $g --no-legs --steps 6 --warmup 2 ) > gpurun_out/profiles_out/r02_sift10m_$g.json 2> gpurun_out/b13_$g.err
echo "sift10m $g rc=$?"; grep -E "build\]|bench\]|real" gpurun_out/b13_$g.err | tail -14
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/profiles_out/r02_sift10m_$g.json').read().strip().splitlines()[-1])
    print("sift10m $g:", d['value'], d['ms_per_step'], d['config']['L'], d['config']['recall_at_10'], d['config']['parity_vs_oracle_first_64'], d['roofline']['avg_launch_us'], d['cpu_baseline'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/b13_$g.err').read()[-2500:])
PY
done
