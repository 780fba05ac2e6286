#!/bin/bash
# Round-end evidence on the GPU box: full -m gpu suite, smoke, the default bench line, rocprofv3 passes of the four single-GPU
# configurations, K2 alone, the shard sweep.  Outputs under gpurun_out/profiles_out/ (copy into profiles/).
set -u
mkdir -p gpurun_out/profiles_out
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/t_final.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/t_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/profiles_out/r02_bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"; tail -2 gpurun_out/bench_default.err
timeout 700 bash tools/profile2.sh r02_sift1m_device sift1m_device --graph device
timeout 700 bash tools/profile2.sh r02_sift1m_host sift1m_host --graph host
PROFILE_PASSES="trace fetch write ea" timeout 500 bash tools/profile2.sh r02_sift1m_host_walker sift1m_host_walker --graph host --pull 0
PROFILE_PASSES="trace fetch write" timeout 600 bash tools/profile2.sh r02_deep100m_shape_device deep100m_shape_device --workload deep100m_shape
PROFILE_PASSES="trace fetch write" timeout 900 bash tools/profile2.sh r02_sift1b_shape_host sift1b_shape_host --workload sift1b_shape
PROFILE_PASSES="trace" timeout 600 bash tools/profile2.sh r02_sift1b_shape_host_walker sift1b_shape_host_walker --workload sift1b_shape --pull 0
timeout 400 bash tools/profile_k2.sh r02 > /dev/null
{
echo "# What one rank's shard of the 10 K-query batch costs on a GPU of its own"
echo
echo "Single-process \`bench.py --queries Q --L 70 --no-legs\` runs (SIFT1M-like index, 6 timed steps each): Q = 10 000 / W queries = the shard of"
echo "rank r of W (host = host graph, rows pulled by the kernel; walker = host graph served by the C++ walker threads)."
echo "No multi-GPU box was available; this is the projection DESIGN.md section 7 quotes."
echo
echo "| graph | queries | QPS | ms per batch | search launch us |"
echo "|---|---|---|---|---|"
for g in device host walker; do for q in 10000 5000 2500 1250; do
  ga="--graph $g"; [ $g = walker ] && ga="--graph host --pull 0"
  timeout 300 python bench.py $ga --queries $q --L 70 --no-legs --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/shard_${g}_$q.json 2> gpurun_out/shard_${g}_$q.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/shard_${g}_$q.json').read().strip().splitlines()[-1])
    print(f"| $g | $q | {d['value']:.0f} | {d['ms_per_step']:.3f} | {d['roofline']['avg_launch_us']:.0f} |")
except Exception as e:
    print("| $g | $q | error | | |")
PY
done; done
} > gpurun_out/profiles_out/r02_shard_sweep.md
cat gpurun_out/profiles_out/r02_shard_sweep.md
ls gpurun_out/profiles_out | wc -l
