#!/bin/bash
# LDS gather micro-benchmark (tools/dev/lds_gather_bench.hip): timings, then the LDS counters of the same launches
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/lds_gather
mkdir -p $OUT
[ -x tools/dev/lds_gather_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/dev/lds_gather_bench tools/dev/lds_gather_bench.hip || exit 1
./tools/dev/lds_gather_bench 200 | tee $OUT/times.jsonl
rm -rf $OUT/pmc
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc -- ./tools/dev/lds_gather_bench 200 > /dev/null 2> $OUT/pmc.err
python3 - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/lds_gather/pmc/**/*counter_collection.csv", recursive=True)
acc = collections.OrderedDict()
for r in csv.DictReader(open(f[0])) if f else []:
    if "k_gather" not in r["Kernel_Name"]:
        continue
    acc.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(acc)
print("| launch (second of each variant) | SQ_INSTS_LDS | SQ_LDS_IDX_ACTIVE | SQ_LDS_BANK_CONFLICT | conflict share | SQ_WAIT_INST_LDS |")
print("|---|---|---|---|---|---|")
for k, name in enumerate(("v0 AoS b64", "v1 SoA 2 x b32", "v2 rotated", "v3 broadcast")):
    if 2 * k + 1 < len(ids):
        c = acc[ids[2 * k + 1]]
        print(f"| {name} | {c.get('SQ_INSTS_LDS', 0):.4g} | {c.get('SQ_LDS_IDX_ACTIVE', 0):.4g} | {c.get('SQ_LDS_BANK_CONFLICT', 0):.4g} | "
              f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 1)):.2f} | {c.get('SQ_WAIT_INST_LDS', 0):.4g} |")
P
