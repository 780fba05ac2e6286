#!/bin/bash
# the three self-paced workloads at HEAD, no side legs: launch time of the search kernel (QPS, ms per batch, launch us, frac, parity)
cd /root/repo
mkdir -p gpurun_out/quick3
for spec in "sift1b_shape 10000" "sift1b_shape 2500" "sift1b_shape 1250" "deep100m_shape 10000" "sift1m 10000" "sift1m 1250"; do
  set -- $spec
  timeout 900 python3 bench.py --workload $1 --queries $2 --no-legs --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/quick3/$1_$2.json 2> gpurun_out/quick3/$1_$2.err
  python3 - gpurun_out/quick3/$1_$2.json "$1 Q=$2" <<'P'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); c=j["config"]; r=j["roofline"]
    print(f"{sys.argv[2]:28s} qps {j['value']:>10} ms {j['ms_per_step']:>8} launch_us {r['avg_launch_us']:>10} frac {r['frac']:.4f} ok", c.get("parity_vs_oracle_first_64", c.get("result_properties_ok")), "recall", c.get("recall_at_10"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
done
