#!/bin/bash
# Are the harness tables reproducible run to run?  (file flow at 1e8 kept on tmpfs, then our harness twice, the reference's, and ours without the fused re-rank)
set -u
O=gpurun_out/r05_det; mkdir -p $O
python tools/file_flow.py --n ${N:-100000000} --out $O/flow --keep --ranks 1 > $O/flow.json 2> $O/flow.err
P=/dev/shm/bang_flow/flow
A="$P ${P}_query.bin ${P}_gt.bin 10000 10 uint8 l2 auto"
B=bang-billion-scale-ann_amd/bin/bang_search
BANG_GRAPH=host $B $A > $O/ours_1.txt 2>&1
BANG_GRAPH=host $B $A > $O/ours_2.txt 2>&1
BANG_GRAPH=host BANG_FUSE_RERANK=0 $B $A > $O/ours_nofuse.txt 2>&1
BANG_GRAPH=host oracle/_ref/ref_bang_search $A > $O/ref_1.txt 2>&1
BANG_GRAPH=host oracle/_ref/ref_bang_search $A > $O/ref_2.txt 2>&1
BANG_GRAPH=device $B $A > $O/ours_device.txt 2>&1
python - <<'PY'
import sys
sys.path.insert(0, "tools")
import file_flow
O = "gpurun_out/r05_det"
t = {k: [(r[0], r[3]) for r in file_flow.table(open(f"{O}/{k}.txt").read())] for k in ("ours_1", "ours_2", "ours_nofuse", "ref_1", "ref_2", "ours_device")}
base = t["ours_1"]
for k, v in t.items():
    diff = [(a, b) for a, b in zip(base, v) if a != b]
    print(k, len(v), "rows; differs from ours_1 in", len(diff), diff[:6])
PY
rm -rf /dev/shm/bang_flow
