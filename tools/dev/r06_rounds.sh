#!/bin/bash
# Two equal rounds for batches of one to two wave-fulls per CU (bang_search_geometry): old library vs new, SIFT1B-shape and SIFT1M-like.
cd "$(dirname "$0")/../.."
TAG=r06_rounds QUERIES=10000,6000,5000,4000,3400,2500 STEPS=8 LIBS="${LIBS:-lib_old lib}" bash tools/dev/r06_ab.sh
TAG=r06_rounds_1m QUERIES=10000,7000,6000,5000 STEPS=8 LIBS="${LIBS:-lib_old lib}" SWEEP_ARGS="--workload sift1m --graph device --L 70" bash tools/dev/r06_ab.sh
