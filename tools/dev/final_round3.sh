#!/bin/bash
# Round-3 evidence on the GPU box: full -m gpu suite, smoke, the default bench line (sift1b_shape + every leg), rocprofv3 passes of
# the single-GPU configurations, K2 alone, the shard sweep on the north-star configuration.  Outputs: gpurun_out/profiles_out/ (-> profiles/).
set -u
R=r03
mkdir -p gpurun_out/profiles_out
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/t_final.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/t_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
s=$(date +%s)
( timeout 1500 python bench.py --steps 20 --warmup 5 ) > gpurun_out/profiles_out/${R}_bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$? wall $(( $(date +%s) - s )) s"; tail -2 gpurun_out/bench_default.err
PROFILE_PASSES="trace fetch write tcp ea" timeout 900 bash tools/profile2.sh ${R}_sift1b_shape_host sift1b_shape_host --workload sift1b_shape
PROFILE_PASSES="trace fetch write tcp ea l2 sq1 sq2" timeout 900 bash tools/profile2.sh ${R}_sift1m_device sift1m_device --workload sift1m --graph device
PROFILE_PASSES="trace fetch write tcp" timeout 700 bash tools/profile2.sh ${R}_sift1m_host sift1m_host --workload sift1m --graph host
PROFILE_PASSES="trace fetch write tcp ea" timeout 700 bash tools/profile2.sh ${R}_deep100m_shape_device deep100m_shape_device --workload deep100m_shape
PROFILE_PASSES="trace" timeout 700 bash tools/profile2.sh ${R}_sift1b_shape_host_walker sift1b_shape_host_walker --workload sift1b_shape --pull 0 --resident-graph
timeout 600 bash tools/profile_k2.sh ${R} > /dev/null
{
echo "# What one rank's shard of the 10 K-query batch costs on a GPU of its own: the north-star configuration"
echo
echo "Single-process \`bench.py --queries Q --no-legs\` runs on the SIFT1B-shape index (streamed load, pull mode, L = 152, 6 timed steps each):"
echo "Q = 10 000 / W queries = the shard of rank r of W.  No multi-GPU box was available; this is the projection DESIGN.md section 7 quotes."
echo
echo "| queries | QPS | ms per batch | search launch us |"
echo "|---|---|---|---|"
for q in 10000 5000 2500 1250; do
  timeout 400 python bench.py --queries $q --no-legs --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/shard_1b_$q.json 2> gpurun_out/shard_1b_$q.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/shard_1b_$q.json').read().strip().splitlines()[-1])
    print(f"| $q | {d['value']:.0f} | {d['ms_per_step']:.3f} | {d['roofline']['avg_launch_us']:.0f} |")
except Exception as e:
    print("| $q | error | | |")
PY
done
} > gpurun_out/profiles_out/${R}_shard_sweep_sift1b.md
cat gpurun_out/profiles_out/${R}_shard_sweep_sift1b.md
./tools/dev/row_fetch_bench > gpurun_out/profiles_out/${R}_row_fetch_bench.jsonl 2>/dev/null
# per-iteration phase times (diagnostic build in lib_prof: make OUT=lib_prof EXTRA_CXXFLAGS=-DBANG_SEARCH_PHASE_PROF)
if [ -f bang-billion-scale-ann_amd/lib_prof/libbang.so ]; then
  { echo "# search_kernel: where an iteration's time goes (s_memrealtime stamps of wave 0 of every workgroup, diagnostic build -DBANG_SEARCH_PHASE_PROF, BANG_SEARCH_PROF=1)"; echo;
    echo '`tools/dev/phase_prof.sh`: bench.py --no-legs --steps 3 per configuration; us per iteration; "Q" = queries in the batch (10 000 = the bench batch, 1 250 = one rank'"'"'s shard of 8).'; echo;
    bash tools/dev/phase_prof.sh 2>&1 | sed 's/^==/\n##/' ; } > gpurun_out/profiles_out/${R}_phase_profile.md
fi
ls gpurun_out/profiles_out | wc -l
