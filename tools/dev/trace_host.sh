#!/bin/bash
# usage: TAG=x [ENV...] tools/dev/trace_host.sh  -- rocprofv3 kernel timeline of the host-graph mode, our kernels only
export TMPDIR=/tmp
OUT=gpurun_out/trace_${TAG:-host}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --L 70 --lanes ${LANES:-4} --no-events ${EXTRA:-} > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/kernels_ours.csv <<'PY'
import csv,sys
r=csv.DictReader(open(sys.argv[1]))
w=csv.writer(sys.stdout)
w.writerow(["kernel","queue","stream","start","end","grid","wg"])
for row in r:
    n=row["Kernel_Name"]
    if any(k in n for k in ("front_kernel","back_kernel","rerank_kernel","center_queries")):
        w.writerow([("front" if "front_kernel" in n else "back" if "back_kernel" in n else n[:20]),row.get("Queue_Id"),row.get("Stream_Id"),row["Start_Timestamp"],row["End_Timestamp"],row.get("Grid_Size_X") or row.get("Grid_Size"),row.get("Workgroup_Size_X") or row.get("Workgroup_Size")])
PY
rm -rf $OUT/t
