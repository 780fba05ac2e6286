#!/bin/bash
# rocprofv3 evidence for one bench configuration (GPU box):   tools/profile2.sh <tag> <traffic-tag> [bench args...]
#   1. kernel trace + stats (per-kernel durations)
#   2. PMC passes, each in its OWN run with --kernel-trace only (MI355X guide: TCC has 4 slots -- FETCH_SIZE takes 3, WRITE_SIZE 2;
#      SQ has 8): HBM traffic, L2 hit/miss/atomics, fabric requests, L1->L2 requests, SQ wait/busy split, LDS conflicts
# Outputs: gpurun_out/profiles_out/<tag>_summary.md, <tag>_kernel_stats.csv, traffic_<traffic-tag>.json  (copy them into profiles/)
set -u
TAG=${1:-r02}; TTAG=${2:-sift1m_device}; shift; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_${TAG}
rm -rf "$OUT"; mkdir -p "$OUT" gpurun_out/profiles_out
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-live-traffic $*"
# Everything that compiles is built BEFORE anything is profiled: the profiler's preloaded library initialises the GPU in every process of
# the tree it starts, and make / hipcc / gcc hops from such a process take the machine down on this pool.  The profiled program is the
# interpreter itself (resolved path, no shim), and BANG_NO_BUILD makes a stale library an error instead of a compile.
python3 -c "import __graft_entry__ as g; g.build(); from oracle import oracle as O; O.build()" || exit 1
PY=$(python3 -c "import os,sys; print(os.path.realpath(sys.executable))")
export BANG_NO_BUILD=1
run_pass() {   # name, rocprof options...
  local name=$1; shift
  rocprofv3 "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- "$PY" bench.py $ARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
}
# PROFILE_PASSES selects the passes (default: all); e.g. PROFILE_PASSES="trace fetch write" for the 240 GB workloads
PASSES=${PROFILE_PASSES:-trace fetch write dram l2 ea tcp sq1 sq2}
for ps in $PASSES; do case $ps in
  trace) run_pass trace --stats ;;
  fetch) run_pass pmc_fetch --pmc FETCH_SIZE ;;
  write) run_pass pmc_write --pmc WRITE_SIZE ;;
  dram) run_pass pmc_dram --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_RDREQ_128B_sum ;;
  l2) run_pass pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum ;;
  ea) run_pass pmc_ea --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_32B_sum ;;
  tcp) run_pass pmc_tcp --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum ;;
  sq1) run_pass pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA ;;
  sq2) run_pass pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU ;;
esac; done
python3 tools/summarize_pmc.py "$OUT" "$TTAG" > "$OUT/summary.md" 2> "$OUT/summary.err"
cp "$OUT/summary.md" gpurun_out/profiles_out/${TAG}_summary.md
cp "$OUT/traffic_${TTAG}.json" gpurun_out/profiles_out/ 2>/dev/null
cp "$OUT/bench_trace.json" gpurun_out/profiles_out/${TAG}_bench_under_trace.json 2>/dev/null
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|search_kernel|front_kernel|back_kernel|rerank_kernel|center_queries|init_state" "$f" > gpurun_out/profiles_out/${TAG}_kernel_stats.csv
tail -5 "$OUT/summary.err"
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
