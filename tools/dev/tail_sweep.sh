#!/bin/bash
# does the last, partly filled round of a 10 000-query batch on 3 072 query slots cost more than its share?  launch time vs batch size
cd /root/repo
mkdir -p gpurun_out/tailsweep
for q in 3072 6144 9216 10000 12288 15360; do
  timeout 600 python3 bench.py --workload sift1b_shape --queries $q --no-legs --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/tailsweep/$q.json 2> gpurun_out/tailsweep/$q.err
  python3 - $q <<'P'
import json,sys
q=sys.argv[1]
try:
    j=json.loads([l for l in open(f"gpurun_out/tailsweep/{q}.json") if l.startswith("{")][-1]); r=j["roofline"]
    print(f"Q={q:>6} launch_us {r['avg_launch_us']:>9} us/query {r['avg_launch_us']/int(q):.4f} qps {j['value']}")
except Exception as e:
    print(q, "FAILED", e)
P
done
