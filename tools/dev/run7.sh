#!/bin/bash
set -u
for numa in 0 1; do
BANG_DEBUG=1 BANG_NUMA=$numa BANG_WALK_PROF=1 timeout 600 python bench.py --graph host --no-legs --no-cpu-baseline --steps 5 --warmup 2 --L 70 > gpurun_out/b7_n$numa.json 2> gpurun_out/b7_n$numa.err
grep -E "pinned|\[walk\]" gpurun_out/b7_n$numa.err | tail -4
python - <<PY
import json
d=json.loads(open('gpurun_out/b7_n$numa.json').read().strip().splitlines()[-1])
print("numa $numa:", d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['step_ms'])
PY
done
cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c | head; numactl -H 2>/dev/null | head -5; grep -E "MemTotal" /sys/devices/system/node/node*/meminfo
