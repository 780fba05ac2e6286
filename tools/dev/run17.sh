#!/bin/bash
set -u
( time timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "pqdist" ) > gpurun_out/t17.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/t17.log
timeout 300 python tools/k2_alone.py --big
