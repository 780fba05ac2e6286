#!/bin/bash
# round 3, GPU call 2: the new default bench line end to end (timed), counter list, waves-per-CU sweep with the filter summary
cd "$(dirname "$0")/../.."
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && rocprofv3 -L > $OLDPWD/$O/counters.txt 2>&1 )
s=$(date +%s)
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "default bench: rc=$? $(( $(date +%s) - s )) s"
tail -c 1500 $O/bench_default.err
python - <<'P'
import json
try:
    j=json.loads(open("gpurun_out/r3a/bench_default.json").read().strip().splitlines()[-1])
    c=j["config"]
    print("value", j["value"], "ms", j["ms_per_step"], "workload", c["workload"][:100])
    print("roofline", {k:v for k,v in j["roofline"].items() if not isinstance(v,(dict,str))})
    print("k2", (j["roofline"].get("k2_alone") or {}).get("frac"))
    print("cpu", j["cpu_baseline"])
    print({k:v for k,v in c.items() if not isinstance(v,(dict,list)) and k!="workload"})
except Exception as e:
    print("parse failed", e)
P
for w in 8 12; do
  BANG_SEARCH_MAX_WAVES=$w python bench.py --workload sift1m --graph device --no-legs --no-cpu-baseline --steps 10 > $O/s1m_w$w.json 2> $O/s1m_w$w.err
  python -c "
import json;j=json.loads(open('$O/s1m_w$w.json').read().strip().splitlines()[-1]);print('sift1m waves $w', j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'])"
done
for w in 8 12; do
  BANG_SEARCH_MAX_WAVES=$w python bench.py --workload sift1b_shape --no-legs --no-cpu-baseline --steps 10 > $O/s1b_w$w.json 2> $O/s1b_w$w.err
  python -c "
import json;j=json.loads(open('$O/s1b_w$w.json').read().strip().splitlines()[-1]);print('sift1b waves $w', j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'])"
done
