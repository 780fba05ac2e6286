#!/bin/bash
# Calibrates the gfx950 memory-side counters on kernels that move a KNOWN number of bytes in the access shapes of the search path
# (tools/traffic_calib.hip): four rocprofv3 --pmc passes (kernel trace only, the binary itself behind `--`), summarised into
# gpurun_out/traffic_calib/summary.md (copied to profiles/r04_traffic_calibration.md).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/traffic_calib
mkdir -p "$OUT" tools/_build
[ -x tools/_build/traffic_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_build/traffic_calib tools/traffic_calib.hip || exit 1
./tools/_build/traffic_calib 16 > "$OUT/known.json" || exit 1
pass() {  # name counters...
  local name=$1; shift
  rm -rf "$OUT/$name"
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- ./tools/_build/traffic_calib 16 > "$OUT/$name.out" 2> "$OUT/$name.err" || echo "pass $name failed (rc $?)"
}
pass p1 FETCH_SIZE TCC_EA0_RDREQ_128B_sum
pass p2 WRITE_SIZE TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_IO_32B_sum
pass p3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass p4 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_BUBBLE_sum
python3 tools/summarize_calibration.py "$OUT" > "$OUT/summary.md"
cat "$OUT/summary.md"
