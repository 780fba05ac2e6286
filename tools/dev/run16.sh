#!/bin/bash
set -u
mkdir -p gpurun_out/profiles_out
( time timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q ) > gpurun_out/t16.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/t16.log
timeout 400 bash tools/profile_k2.sh r02 > /dev/null
head -12 gpurun_out/profiles_out/r02_k2_alone.md
( time timeout 900 python bench.py --steps 10 --warmup 3 ) > gpurun_out/profiles_out/r02_bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"; tail -4 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/profiles_out/r02_bench_default.json').read().strip().splitlines()[-1])
c=d['config']
print('value', d['value'], d['ms_per_step'], c['graph'], c['graph_placement'], 'frac', d['roofline']['frac'])
for k in c:
    if k.startswith('at_'):
        v=c[k]
        print(k, {kk: v.get(kk) for kk in ('graph','L','queries_per_s','ms_per_batch','recall_at_10','result_properties_ok','parity_vs_oracle_first_64','error')})
print('k2', d['roofline'].get('k2_alone'))
print('cpu', d['cpu_baseline'])
PY
