#!/bin/bash
# N > 1 logic of bench.py on a 1-GPU box: W ranks launched as the driver launches them (torch.distributed.run) and -- round 6 -- by bench.py
# itself (plain `python bench.py --gpus W`: tools/bench_legs/launch.py), all on GPU 0 (BANG_BENCH_SHARE_GPU), ONE host graph / ONE pull-rows
# file for the node, the job's single gather over gloo (RCCL refuses two ranks on one device).  The 8-rank line exercises the slice table at 8
# of BANG_MAX_ROW_SLICES = 16 (peer rows over hipIpc among eight processes).
# The ranks share the CUs of one GPU, so these lines show that the sharded job is correct -- not how it scales.
set -u
OUT=gpurun_out/profiles_out/${TAG:-r03}_dryrun_shared_gpu.jsonl
mkdir -p gpurun_out/profiles_out; : > $OUT
export BANG_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1
port=29611
for g in device host; do for w in 1 2 4; do
  args="bench.py --gpus $w --workload sift1m --graph $g --L 70 --steps 6 --warmup 2 --no-cpu-baseline --no-legs --backend gloo"
  if [ $w = 1 ]; then timeout 600 python $args 2> gpurun_out/dry_${g}_$w.err | tail -1 >> $OUT
  else port=$((port+1)); timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $port $args 2> gpurun_out/dry_${g}_$w.err | grep '^{' | tail -1 >> $OUT; fi
done; done
port=$((port+1)); timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --workload sift1m --graph device --L 70 --steps 6 --warmup 2 --no-cpu-baseline --no-legs --backend gloo --batches 2 2> gpurun_out/dry_weak.err | grep '^{' | tail -1 >> $OUT
port=$((port+1)); timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --workload sift1b_shape --shape-n 12000000 --steps 4 --warmup 1 --no-cpu-baseline --no-legs --backend gloo 2> gpurun_out/dry_1b.err | grep '^{' | tail -1 >> $OUT
# the ranks started by bench.py itself: 4 and 8 ranks on the north-star layout (streamed SIFT1B-shape index at reduced N, rows pulled / peer rows)
for w in 4 8; do
  timeout 1200 python bench.py --gpus $w --workload sift1b_shape --shape-n 12000000 --queries 2048 --L 40 --steps 4 --warmup 1 --no-cpu-baseline --no-legs --backend gloo 2> gpurun_out/dry_self_$w.err | grep '^{' | tail -1 >> $OUT
done
python - <<'PY'
import json
import os
for l in open(f"gpurun_out/profiles_out/{os.environ.get('TAG', 'r03')}_dryrun_shared_gpu.jsonl"):
    try:
        d=json.loads(l); c=d['config']
        print(d['n_gpus'], d.get('ranks_started_by'), d.get('world_seen'), c['graph'], d['scaling'], round(d['value']), d['ms_per_step'], c.get('parity_vs_oracle_first_64', c.get('result_properties_ok')), c.get('recall_at_10'), (c.get('peer_rows') or {}).get('fraction'), d.get('peer_rows_fallback'), c['host_loop'][:50])
    except Exception as e:
        print('bad line', e)
PY
