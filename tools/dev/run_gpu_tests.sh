#!/bin/bash
# GPU box: full -m gpu suite, then the default bench line (with legs), then the counter list of rocprofv3.
set -u
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/t_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_gpu.log
tail -5 gpurun_out/t_gpu.log
( time python bench.py --steps 10 --warmup 3 ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?"
tail -3 gpurun_out/bench_default.err
head -c 3000 gpurun_out/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/gpurun_out/counters.txt
