#!/bin/bash
set -u
# A/B: libraries under bang-billion-scale-ann_amd/lib_ab/libbang_<v>.so (BANG_AMD_LIB override)
VARIANTS="${VARIANTS:-v0 v1}"
run() {  # label, args...
  local label=$1; shift
  timeout 600 python bench.py "$@" --no-legs --no-cpu-baseline > gpurun_out/b20_$label.json 2> gpurun_out/b20_$label.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b20_$label.json').read().strip().splitlines()[-1])
    print("$label:", round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'], d.get('result_properties_ok', d['config'].get('result_properties_ok')))
except Exception as e:
    print("$label error", e)
PY
}
for v in $VARIANTS; do
  export BANG_AMD_LIB=$PWD/bang-billion-scale-ann_amd/lib_ab/libbang_$v.so
  timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "search or host" 2>&1 | tail -1
  for q in 10000 2500 1250; do
    run ${v}_device_$q --graph device --queries $q --L 70 --steps 6 --warmup 2
    run ${v}_host_$q --graph host --queries $q --L 70 --steps 6 --warmup 2
  done
  run ${v}_deep --workload deep100m_shape --steps 5 --warmup 2
  run ${v}_sift1b --workload sift1b_shape --steps 5 --warmup 2
done
