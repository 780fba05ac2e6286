#!/bin/bash
set -u
# A/B of two libraries (lib_ab/libbang_<v>.so): K2 alone + the main bench configurations
run() {  # label, args...
  local label=$1; shift
  timeout 900 python bench.py "$@" --no-legs --no-cpu-baseline > gpurun_out/b28_$label.json 2> gpurun_out/b28_$label.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/b28_$label.json').read().strip().splitlines()[-1])
    c=d['config']
    print("$label:", round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'], c.get('parity_vs_oracle_first_64', c.get('result_properties_ok')))
except Exception as e:
    print("$label error", e)
PY
}
for v in ${VARIANTS:-base noslp}; do
  export BANG_AMD_LIB=$PWD/bang-billion-scale-ann_amd/lib_ab/libbang_$v.so
  timeout 300 python tools/k2_alone.py --big 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l)
        print('$v k2 m=%s: %.1f us (min %.1f) %.1f GB/s %.2f G rows/s' % (d.get('m'), d.get('avg_launch_us', 0), d.get('min_launch_us', 0), d.get('achieved', 0), d.get('rows_per_s', 0)))
    except Exception:
        pass
"
  run ${v}_device --graph device --L 70 --steps 10 --warmup 3
  run ${v}_device_1250 --graph device --L 70 --queries 1250 --steps 10 --warmup 3
  run ${v}_pull --graph host --L 70 --steps 10 --warmup 3
  run ${v}_walker --graph host --pull 0 --L 70 --steps 10 --warmup 3
  run ${v}_deep --workload deep100m_shape --steps 5 --warmup 2
  run ${v}_sift1b --workload sift1b_shape --steps 5 --warmup 2
  run ${v}_sift1b_walker --workload sift1b_shape --shape-n 371000000 --pull 0 --steps 5 --warmup 2
done
