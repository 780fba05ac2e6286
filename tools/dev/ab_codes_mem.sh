#!/bin/bash
# A/B: the PQ code table of the SIFT1B-shape / DEEP100M-shape workloads in ordinary device memory vs hipDeviceMallocUncached (0x3) vs
# fine-grained (0x1): do code rows that are never re-read stop evicting the visited filters from L2 / the Infinity Cache?
cd /root/repo
mkdir -p gpurun_out/abmem
for wl in deep100m_shape sift1b_shape; do
  for fl in ${FLAGS:-"" 0x3 0x1}; do
    tag=${wl}_${fl:-plain}
    SHAPE_CODES_MEMFLAGS=$fl timeout 900 python3 bench.py --workload $wl --no-legs --steps 6 --warmup 2 > gpurun_out/abmem/$tag.json 2> gpurun_out/abmem/$tag.err
    python3 - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads([l for l in open(f"gpurun_out/abmem/{t}.json") if l.startswith("{")][-1])
    print(t, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"].get("k2_alone", {}).get("avg_launch_us"))
except Exception as e:
    print(t, "FAILED", e)
PY
  done
done
