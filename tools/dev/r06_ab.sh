#!/bin/bash
# Round 6 A/B: one shard sweep per library (LIBS), all cells of a library on ONE engine load; optional DEEP100M-shape bench lines (DEEP=1).
cd "$(dirname "$0")/../.."
O=gpurun_out/${TAG:-r06_ab}; mkdir -p $O
PKG=$PWD/bang-billion-scale-ann_amd
if [ -n "${TESTS:-}" ]; then timeout 1500 python -m pytest $TESTS -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log; fi
for lib in ${LIBS:-lib}; do
  [ -f $PKG/$lib/libbang.so ] || { echo "no $lib"; continue; }
  if [ "${SWEEP:-1}" = 1 ]; then
    BANG_AMD_LIB=$PKG/$lib/libbang.so BANG_NO_BUILD=1 timeout 1500 python tools/shard_sweep.py --queries ${QUERIES:-10000,2500,1250} --variants ${VARIANTS:-default} --steps ${STEPS:-8} --check \
      ${SWEEP_ARGS:-} --out $O/sweep_$lib.md > $O/sweep_$lib.log 2> $O/sweep_$lib.err
    echo "== $lib"; grep '^| [0-9]' $O/sweep_$lib.md
  fi
  if [ "${SIFT1M:-0}" = 1 ]; then
    BANG_AMD_LIB=$PKG/$lib/libbang.so BANG_NO_BUILD=1 timeout 900 python bench.py --workload sift1m --graph device --L 70 --no-legs --no-cpu-baseline --no-live-traffic --steps 10 --warmup 3 > $O/sift1m_$lib.json 2> $O/sift1m_$lib.err
    python - $O/sift1m_$lib.json <<'P'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j["config"]; r=j["roofline"]
    print("sift1m qps", j["value"], "ms", j["ms_per_step"], "launch_us", r["avg_launch_us"], "recall", c.get("recall_at_10"), "ok", c.get("parity_vs_oracle_first_64"))
except Exception as e:
    print("sift1m FAILED", e)
P
  fi
  if [ "${DEEP:-0}" = 1 ]; then
    BANG_AMD_LIB=$PKG/$lib/libbang.so BANG_NO_BUILD=1 timeout 900 python bench.py --workload deep100m_shape --no-legs --no-cpu-baseline --no-live-traffic --steps 10 --warmup 3 > $O/deep_$lib.json 2> $O/deep_$lib.err
    python - $O/deep_$lib.json <<'P'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=j["config"]; r=j["roofline"]
    print("deep100m_shape qps", j["value"], "ms", j["ms_per_step"], "launch_us", r["avg_launch_us"], "fused", c["rerank_fused"], "ok", c.get("result_properties_ok"))
except Exception as e:
    print("deep FAILED", e)
P
  fi
done
