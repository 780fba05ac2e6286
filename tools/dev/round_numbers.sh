#!/bin/bash
# Collects the numbers quoted in DESIGN.md section 6 (one bench.py run per line) into gpurun_out/numbers/.
mkdir -p gpurun_out/numbers
run() { name=$1; shift; env "$@" timeout 900 python bench.py ${BARGS} 2>gpurun_out/numbers/$name.err | tail -1 > gpurun_out/numbers/$name.json; python3 - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/numbers/{n}.json")); c = d["config"]
    print(n, d["value"], "q/s", d["ms_per_step"], "ms L", c["L"], "recall", c["recall_at_10"], "it", c["iterations"], c["host_loop"], "|", c["rerank_vectors"],
          "| minmax", c["step_ms_min_max"], "| parity", c.get("parity_vs_oracle_first_64", c.get("result_properties_ok")), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(n, "FAILED", e)
PY
}
BARGS="--steps 10 --warmup 3" run host_default X=1
BARGS="--steps 10 --warmup 3 --no-cpu-baseline --graph device" run device_default X=1
BARGS="--steps 10 --warmup 3 --no-cpu-baseline" run host_vectors_shipped BANG_VECTORS=0
BARGS="--steps 10 --warmup 3 --no-cpu-baseline" run host_launch_per_iter BANG_PERSISTENT=0 BANG_VECTORS=0
BARGS="--steps 10 --warmup 3 --no-cpu-baseline --graph device" run device_launch_per_iter BANG_PERSISTENT=0
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --workload deep100m_shape --graph device" run deep100m_shape_device X=1
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --workload deep100m_shape --graph device" run deep100m_shape_device_launch_per_iter BANG_PERSISTENT=0
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --workload sift1b_shape" run sift1b_shape_host X=1
BARGS="--steps 5 --warmup 2 --no-cpu-baseline --workload sift1b_shape" run sift1b_shape_host_launch_per_iter BANG_PERSISTENT=0
