#!/bin/bash
set -u
BANG_WALK_PROF=1 timeout 600 python bench.py --graph host --no-legs --no-cpu-baseline --steps 3 --warmup 2 --L 70 > gpurun_out/b6.json 2> gpurun_out/b6.err
grep "\[walk\]" gpurun_out/b6.err | tail -12
python - <<PY
import json
d=json.loads(open('gpurun_out/b6.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['walker_threads'])
PY
for t in 6 14; do
BANG_WALK_PROF=1 timeout 600 python bench.py --graph host --no-legs --no-cpu-baseline --steps 3 --warmup 2 --L 70 --threads $t > gpurun_out/b6_t$t.json 2> gpurun_out/b6_t$t.err
grep "\[walk\]" gpurun_out/b6_t$t.err | tail -3
python - <<PY
import json
d=json.loads(open('gpurun_out/b6_t$t.json').read().strip().splitlines()[-1])
print("threads $t:", d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['walker_threads'])
PY
done
lscpu | grep -E "Model name|Socket|NUMA|^CPU\(s\)"; cat /sys/fs/cgroup/cpu.max; nproc
