#!/bin/bash
# A/B: PQ code rows requested with the non-temporal hint (lib_nt) vs default cache policy
cd "$(dirname "$0")/../.."
O=gpurun_out/ab_nt; mkdir -p $O
NT=$PWD/bang-billion-scale-ann_amd/lib_nt/libbang.so
run() { local name=$1; shift; local envs=$1; shift
  env $envs timeout 900 python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'P'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j["config"]; r=j["roofline"]
    print(sys.argv[2], "qps", j["value"], "ms", j["ms_per_step"], "launch_us", r["avg_launch_us"], "frac", r["frac"], "ok", c.get("parity_vs_oracle_first_64", c.get("result_properties_ok")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
run sift1m_dev "X=1" --workload sift1m --graph device
run sift1m_dev_nt "BANG_AMD_LIB=$NT" --workload sift1m --graph device
run sift1m_host "X=1" --workload sift1m --graph host
run sift1m_host_nt "BANG_AMD_LIB=$NT" --workload sift1m --graph host
run deep "X=1" --workload deep100m_shape
run deep_nt "BANG_AMD_LIB=$NT" --workload deep100m_shape
run sift1b "X=1" --workload sift1b_shape
run sift1b_nt "BANG_AMD_LIB=$NT" --workload sift1b_shape
run sift10m_nt "BANG_AMD_LIB=$NT" --workload sift10m --graph device
run sift10m "X=1" --workload sift10m --graph device
