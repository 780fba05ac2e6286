#!/bin/bash
# What one rank's shard of the 10 K-query batch costs on the SIFT1B-shape index (streamed load, pull mode), a GPU of its own
set -u
mkdir -p gpurun_out/profiles_out
OUT=gpurun_out/profiles_out/r02_shard_sweep_sift1b.md
{
echo "# SIFT1B-shape (streamed load, N = 0.94e9, L = 152, rows pulled by the kernel): one rank's shard of the 10 K-query batch on a GPU of its own"
echo
echo "Single-process \`bench.py --workload sift1b_shape --queries Q --no-legs\` runs, 5 timed steps each; Q = 10 000 / W."
echo
echo "| queries | QPS | ms per batch | search launch us | result properties |"
echo "|---|---|---|---|---|"
for q in 10000 5000 2500 1250; do
  timeout 600 python bench.py --workload sift1b_shape --queries $q --no-legs --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/ss1b_$q.json 2> gpurun_out/ss1b_$q.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/ss1b_$q.json').read().strip().splitlines()[-1])
    print(f"| $q | {d['value']:.0f} | {d['ms_per_step']:.3f} | {d['roofline']['avg_launch_us']:.0f} | {d['config'].get('result_properties_ok')} |")
except Exception as e:
    print("| $q | error | | | |")
PY
done
} > $OUT
cat $OUT
