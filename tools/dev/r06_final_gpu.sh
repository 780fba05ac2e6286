#!/bin/bash
# Round 6, final records on the final code: the driver-style bench line with every leg, the rocprofv3 passes of the three BASELINE configurations,
# the multi-rank dry run (1 / 2 / 4 ranks under the launcher, 4 / 8 ranks started by bench.py itself), a 10-minute random-configuration soak.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06f; mkdir -p $O gpurun_out/profiles_out
(time python bench.py --steps 20 --warmup 5) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
PROFILE_PASSES="trace fetch write dram l2 ea sq1 sq2" bash tools/profile2.sh r06_sift1b_shape_host sift1b_shape_host > $O/profile2_sift1b.log 2>&1; tail -2 $O/profile2_sift1b.log
PROFILE_PASSES="trace fetch write dram l2 ea sq1" bash tools/profile2.sh r06_sift1m_device sift1m_device --workload sift1m --graph device --L 70 > $O/profile2_sift1m.log 2>&1; tail -2 $O/profile2_sift1m.log
PROFILE_PASSES="trace fetch write dram l2 ea sq1" bash tools/profile2.sh r06_deep100m_shape_device deep100m_shape_device --workload deep100m_shape > $O/profile2_deep.log 2>&1; tail -2 $O/profile2_deep.log
TAG=r06 bash tools/dev/dryrun_shared_gpu.sh > $O/dryrun.log 2>&1; tail -14 $O/dryrun.log
timeout 900 python tools/soak_random.py 3000 20261006 600 > $O/soak.log 2>&1; tail -3 $O/soak.log
ls gpurun_out/profiles_out
