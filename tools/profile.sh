#!/bin/bash
# Collects the rocprofv3 evidence on the GPU box:  tools/profile.sh <tag> <traffic-tag> [bench args...]
#   1. kernel trace + stats of the bench command (per-kernel durations)
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -- counters in their own runs, as the MI355X guide prescribes
# Outputs: gpurun_out/prof_<tag>/summary.md (+ kernel_stats.csv) and profiles/traffic_<traffic-tag>.json; copy both into profiles/.
set -u
TAG=${1:-r01}; TTAG=${2:-sift1m_host}; shift; shift || true
export TMPDIR=/tmp
export BANG_BENCH_NO_L200=1     # keep the timed steps the LAST launches of the run (tools/summarize_profile.py picks them by position)
OUT=gpurun_out/prof_${TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
python3 -c "import __graft_entry__ as g; g.build()" || exit 1     # build BEFORE profiling: no compiler may start under rocprofv3 --pmc
PY=$(python3 -c "import os,sys; print(os.path.realpath(sys.executable))")
export BANG_NO_BUILD=1
export BANG_BENCH_NO_TRAFFIC=1   # a profiled bench must never start its own nested rocprofv3 passes (bench.py: live_traffic())
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$PY" bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- "$PY" bench.py $ARGS > "$OUT/bench_pmc_fetch.json" 2> "$OUT/bench_pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- "$PY" bench.py $ARGS > "$OUT/bench_pmc_write.json" 2> "$OUT/bench_pmc_write.err"
python3 tools/summarize_profile.py "$OUT" "$TTAG" > "$OUT/summary.md" 2> "$OUT/summary.err"
mkdir -p gpurun_out/profiles_out
cp "$OUT/summary.md" gpurun_out/profiles_out/${TAG}_summary.md
cp profiles/traffic_${TTAG}.json gpurun_out/profiles_out/ 2>/dev/null
cp "$OUT/bench_trace.json" gpurun_out/profiles_out/${TAG}_bench_under_trace.json 2>/dev/null
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|front_kernel|back_kernel|rerank_kernel|center_queries|init_state" "$f" > gpurun_out/profiles_out/${TAG}_kernel_stats.csv
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write"
