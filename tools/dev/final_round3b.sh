#!/bin/bash
# refresh of the round-3 evidence at HEAD: full -m gpu suite, smoke, default bench line, rocprofv3 passes of the bench line's launch
set -u
R=r03
mkdir -p gpurun_out/profiles_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/t_final.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/t_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
s=$(date +%s)
( timeout 1500 python bench.py --steps 20 --warmup 5 ) > gpurun_out/profiles_out/${R}_bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$? wall $(( $(date +%s) - s )) s"
PROFILE_PASSES="trace fetch write tcp ea" timeout 900 bash tools/profile2.sh ${R}_sift1b_shape_host sift1b_shape_host --workload sift1b_shape
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
