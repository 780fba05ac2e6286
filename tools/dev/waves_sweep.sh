#!/bin/bash
# the 10 K batch at 8 / 10 / 11 / 12 waves per CU (BANG_SEARCH_MAX_WAVES): fewer waves = shorter query lifetime = shorter drain
cd /root/repo
mkdir -p gpurun_out/wavesweep
for wl in sift1b_shape deep100m_shape; do
for w in 8 10 11 12; do
  BANG_SEARCH_MAX_WAVES=$w timeout 600 python3 bench.py --workload $wl --no-legs --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/wavesweep/${wl}_$w.json 2> gpurun_out/wavesweep/${wl}_$w.err
  python3 - $wl $w <<'P'
import json,sys
wl,w=sys.argv[1:3]
try:
    j=json.loads([l for l in open(f"gpurun_out/wavesweep/{wl}_{w}.json") if l.startswith("{")][-1]); r=j["roofline"]
    print(f"{wl} waves={w:>3} launch_us {r['avg_launch_us']:>9} qps {j['value']}")
except Exception as e:
    print(wl, w, "FAILED", e)
P
done; done
