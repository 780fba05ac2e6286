#!/bin/bash
# A/B of the on-chip filter summary (search kernel): default build vs -DBANG_FILTER_SUMMARY=0 (lib_nosumm), three workloads
cd "$(dirname "$0")/../.."
O=gpurun_out/ab_summary; mkdir -p $O
NOS=$PWD/bang-billion-scale-ann_amd/lib_nosumm/libbang.so
run() { # name, env, args
  local name=$1; shift; local envs=$1; shift
  env $envs timeout 900 python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'P'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j["config"]; r=j["roofline"]
    print(sys.argv[2], "qps", j["value"], "ms", j["ms_per_step"], "launch_us", r["avg_launch_us"], "frac", r["frac"], "probes", c.get("filter_probes_per_step"), "skipped", c.get("filter_loads_skipped_per_step"), "recall", c.get("recall_at_10"), "ok", c.get("parity_vs_oracle_first_64", c.get("result_properties_ok")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
for rep in 1 2; do
run sift1m_dev_summ_$rep "X=1" --workload sift1m --graph device
run sift1m_dev_nosumm_$rep "BANG_AMD_LIB=$NOS" --workload sift1m --graph device
run sift1m_host_summ_$rep "X=1" --workload sift1m --graph host
run sift1m_host_nosumm_$rep "BANG_AMD_LIB=$NOS" --workload sift1m --graph host
done
run deep_summ "X=1" --workload deep100m_shape
run deep_nosumm "BANG_AMD_LIB=$NOS" --workload deep100m_shape
run sift1b_summ "X=1" --workload sift1b_shape
run sift1b_nosumm "BANG_AMD_LIB=$NOS" --workload sift1b_shape
