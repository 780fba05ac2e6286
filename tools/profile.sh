#!/bin/bash
# Collects the rocprofv3 evidence for a round on the GPU box:  tools/profile.sh <tag> [bench args...]
#   1. kernel trace + stats of the bench command (per-kernel durations)
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -- counters in their own runs, as the MI355X guide prescribes
# Outputs land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_pmc_fetch.json" 2> "$OUT/bench_pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_pmc_write.json" 2> "$OUT/bench_pmc_write.err"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summary.err"
# keep the merged-back payload small: drop the per-dispatch traces, keep stats + summary
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
ls -R "$OUT" | head -50
