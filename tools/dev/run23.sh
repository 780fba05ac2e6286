#!/bin/bash
set -u
run() {  # label, args...
  local label=$1; shift
  timeout 900 python bench.py "$@" --no-legs --no-cpu-baseline > gpurun_out/b23_$label.json 2> gpurun_out/b23_$label.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b23_$label.json').read().strip().splitlines()[-1])
    c=d['config']
    print("$label:", round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'], c.get('parity_vs_oracle_first_64'), c.get('result_properties_ok'), c.get('recall_at_10'))
except Exception as e:
    print("$label error", e)
PY
}
BANG_PULL=1 timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -2
for pull in 1 0; do
  export BANG_PULL=$pull
  for q in 10000 2500 1250; do run pull${pull}_host_$q --graph host --queries $q --L 70 --steps 6 --warmup 2; done
  run pull${pull}_sift1b --workload sift1b_shape --shape-n 370000000 --steps 5 --warmup 2
  grep -h "pull rows\|\[bang\] alloc" gpurun_out/b23_pull${pull}_sift1b.err | tail -2
done
