#!/bin/bash
# instruction-cache counters of the search launch for two libraries (LIBS), 10 K SIFT1B-shape batch at N = 2e8 (fast to load)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/${TAG:-r06_icache}; rm -rf $O; mkdir -p $O
PKG=$PWD/bang-billion-scale-ann_amd
python3 -c "import __graft_entry__ as g; g.build(); from oracle import oracle as O; O.build()" || exit 1
PY=$(python3 -c "import os,sys; print(os.path.realpath(sys.executable))")
export BANG_NO_BUILD=1 BANG_BENCH_NO_TRAFFIC=1
for lib in $LIBS; do
  export BANG_AMD_LIB=$PKG/$lib/libbang.so
  for pass in "ic:SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH SQ_WAIT_ANY"; do
    n=${pass%%:*}; c=${pass#*:}
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${lib}_$n -- "$PY" bench.py --shape-n 200000000 --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-live-traffic > $O/${lib}_$n.json 2> $O/${lib}_$n.err
    python3 - $O/${lib}_$n $lib $n <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(sys.argv[2], sys.argv[3], "no counters"); sys.exit()
rows = [r for r in csv.DictReader(open(f[0])) if "search_kernel" in r["Kernel_Name"]]
ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-3:]
agg = {}
for r in rows:
    if int(r["Dispatch_Id"]) in ids:
        agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / len(ids)
print(sys.argv[2], sys.argv[3], {k: f"{v:.4g}" for k, v in sorted(agg.items())})
P
    rm -rf $O/${lib}_$n
  done
done
