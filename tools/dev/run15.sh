#!/bin/bash
set -u
mkdir -p gpurun_out/profiles_out
timeout 600 bash tools/profile2.sh r02_sift1m_host sift1m_host --graph host
PROFILE_PASSES="trace fetch write" timeout 600 bash tools/profile2.sh r02_deep100m_shape_device deep100m_shape_device --workload deep100m_shape
PROFILE_PASSES="trace fetch write" timeout 900 bash tools/profile2.sh r02_sift1b_shape_host sift1b_shape_host --workload sift1b_shape
timeout 400 bash tools/profile_k2.sh r02 > /dev/null
timeout 200 tools/dev/rand_sector_bench > gpurun_out/profiles_out/r02_rand_sector_bench.jsonl 2>&1
ls -la gpurun_out/profiles_out
