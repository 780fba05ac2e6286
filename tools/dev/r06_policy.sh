#!/bin/bash
# Round 6: launch policies re-measured on the final kernel (summary cut-off, spec_rows) and the peer-rows sensitivity (rows_frac at N = 5e8)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_policy; mkdir -p $O

timeout 1200 python tools/shard_sweep.py --queries 10000,2500,1900,1536,1250 --variants BANG_SUMM_ITERS=-1,BANG_SUMM_ITERS=1 --steps 6 --check --out $O/summ.md > $O/summ.log 2> $O/summ.err; grep '^| [0-9]' $O/summ.md
timeout 1200 python tools/shard_sweep.py --queries 10000,2500,1250 --variants BANG_SPEC_ROWS=1,BANG_SPEC_ROWS=2 --steps 6 --check --out $O/spec.md > $O/spec.log 2> $O/spec.err; grep '^| [0-9]' $O/spec.md
timeout 1200 python tools/shard_sweep.py --shape-n 500000000 --queries 10000,2500,1250 --variants rows_frac=0.18,rows_frac=0.36,rows_frac=0.71,rows_frac=1.0 --steps 6 --check --out $O/rows_frac.md > $O/rows_frac.log 2> $O/rows_frac.err; grep '^| [0-9]' $O/rows_frac.md
