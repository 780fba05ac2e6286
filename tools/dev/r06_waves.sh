#!/bin/bash
# Round 6: the long-row search instances compiled for 16 waves (128 VGPRs, 144 words of scratch per wave) against the 12-wave build.
# Parity first (the kernel-level, policy and engine tests on the new default library), then one shard sweep per library with the
# waves per CU capped at 12 / 14 / 16 (BANG_SEARCH_MAX_WAVES) -- all cells of one library on ONE engine load, ids checked across the variants.
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_waves; mkdir -p $O
PKG=$PWD/bang-billion-scale-ann_amd
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_policies.py tests/test_gpu_engine.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
fi
for lib in ${LIBS:-lib lib_768 lib_q0 lib_768f}; do
  [ -f $PKG/$lib/libbang.so ] || continue
  vars="BANG_SEARCH_MAX_WAVES=12,BANG_SEARCH_MAX_WAVES=14,BANG_SEARCH_MAX_WAVES=16"
  case $lib in lib_768*) vars="default";; esac
  BANG_AMD_LIB=$PKG/$lib/libbang.so BANG_NO_BUILD=1 timeout 1500 python tools/shard_sweep.py --queries ${QUERIES:-10000,2500,1250} --variants $vars --steps ${STEPS:-6} --check \
    --out $O/sweep_$lib.md > $O/sweep_$lib.log 2> $O/sweep_$lib.err
  echo "== $lib"; grep '^|' $O/sweep_$lib.md
done
