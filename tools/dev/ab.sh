#!/bin/bash
# generic A/B:  tools/dev/ab.sh <variant lib dir name (under bang-billion-scale-ann_amd/)> <workload spec>...
#   workload spec = "name:bench args", e.g. "sift1m_dev:--workload sift1m --graph device"
cd "$(dirname "$0")/../.."
V=$1; shift
O=gpurun_out/ab_$V; mkdir -p $O
ALT=$PWD/bang-billion-scale-ann_amd/$V/libbang.so
run() { local name=$1; shift; local envs=$1; shift
  env $envs timeout 900 python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 $@ > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'P'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j["config"]; r=j["roofline"]
    print(f"{sys.argv[2]:28s} qps {j['value']:>10} ms {j['ms_per_step']:>8} launch_us {r['avg_launch_us']:>10} frac {r['frac']:.4f} ok", c.get("parity_vs_oracle_first_64", c.get("result_properties_ok")), "recall", c.get("recall_at_10"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  run ${name} "X=1" $args
  run ${name}_$V "BANG_AMD_LIB=$ALT" $args
done
