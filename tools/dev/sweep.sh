#!/bin/bash
# usage: tools/dev/sweep.sh "<env assignments>" ...   -- runs bench.py once per argument and prints value / ms_per_step
for cfg in "$@"; do
  out=$(env $cfg timeout 250 python bench.py --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1)
  echo "$cfg => $(echo "$out" | python3 -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); c=d["config"]; print(round(d["value"]), "q/s", d["ms_per_step"], "ms", "L=",c.get("L"), "it", c.get("iterations"), "front", c.get("front_ms_per_step"), "busy", c.get("front_busy_ms_per_step"), "walk", c.get("walker_ms_per_step"), "sync", c.get("sync_ms_per_step"), "enq", c.get("enqueue_ms_per_step"), "minmax", c.get("step_ms_min_max"))
except Exception as e: print("ERR", e)')"
done
