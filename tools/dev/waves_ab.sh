#!/bin/bash
# 11 vs 12 waves per CU on the bench configuration, alternating, same box
cd /root/repo
mkdir -p gpurun_out/wavesab
for rep in 1 2 3; do
for w in 12 11; do
  BANG_SEARCH_MAX_WAVES=$w timeout 600 python3 bench.py --workload sift1b_shape --no-legs --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/wavesab/${w}_$rep.json 2> gpurun_out/wavesab/${w}_$rep.err
  python3 - $w $rep <<'P'
import json,sys
w,rep=sys.argv[1:3]
j=json.loads([l for l in open(f"gpurun_out/wavesab/{w}_{rep}.json") if l.startswith("{")][-1]); r=j["roofline"]
print(f"waves={w:>3} rep {rep} launch_us {r['avg_launch_us']:>9} ms/step {j['ms_per_step']} qps {j['value']}")
P
done; done
for q in 5000 2500 7000 12000; do for w in 12 11 10; do
  BANG_SEARCH_MAX_WAVES=$w timeout 600 python3 bench.py --workload sift1b_shape --queries $q --no-legs --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/wavesab/q${q}_$w.json 2> /dev/null
  python3 - $q $w <<'P'
import json,sys
q,w=sys.argv[1:3]
j=json.loads([l for l in open(f"gpurun_out/wavesab/q{q}_{w}.json") if l.startswith("{")][-1]); r=j["roofline"]
print(f"Q={q:>6} waves={w:>3} launch_us {r['avg_launch_us']:>9}")
P
done; done
