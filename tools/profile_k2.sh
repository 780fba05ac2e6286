#!/bin/bash
# rocprofv3 evidence for the PQ-distance stage ALONE (K2, compute_neighborDist_par, bang_search.cu:1201-1241):
#   tools/k2_alone.py --big = bang_k_pqdist_stream over 40 M (query, neighbour) pairs per launch on a 4 GB code table, m = 32 / 70 / 74
#   (packed and 128-byte rows): 16 launches per layout, the first 6 are warm-up.
# Pass 1: kernel trace + stats; pass 2: FETCH_SIZE; pass 3: L2 hit / miss / requests; passes 4-5: SQ wait/busy split, instruction mix, LDS conflicts.  Output: gpurun_out/profiles_out/<tag>_k2_alone.md
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_k2
rm -rf "$OUT"; mkdir -p "$OUT" gpurun_out/profiles_out
python3 -c "import __graft_entry__ as g; g.build()" || exit 1     # build BEFORE profiling: no compiler may start under rocprofv3 --pmc
PY=$(python3 -c "import os,sys; print(os.path.realpath(sys.executable))")
export BANG_NO_BUILD=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$PY" tools/k2_alone.py --big > "$OUT/k2_trace.jsonl" 2> "$OUT/k2_trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- "$PY" tools/k2_alone.py --big > "$OUT/k2_fetch.jsonl" 2> "$OUT/k2_fetch.err"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d "$OUT/pmc_l2" -- "$PY" tools/k2_alone.py --big > "$OUT/k2_l2.jsonl" 2> "$OUT/k2_l2.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d "$OUT/pmc_sq1" -- "$PY" tools/k2_alone.py --big > "$OUT/k2_sq1.jsonl" 2> "$OUT/k2_sq1.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- "$PY" tools/k2_alone.py --big > "$OUT/k2_sq2.jsonl" 2> "$OUT/k2_sq2.err"
python3 - "$OUT" > gpurun_out/profiles_out/${TAG}_k2_alone.md <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
NL, WARM = 16, 6          # launches per layout in tools/k2_alone.py --big (bench.k2_alone: 6 warm-up + reps = 10 timed)
LAYOUTS = ("32", "70", "70 (rows 128 B apart)", "74", "74 (rows 128 B apart)")
print("# K2 alone (bang_k_pqdist_stream: pqdist_stream_kernel) under rocprofv3\n")
print("`tools/k2_alone.py --big`: 40 M (query, neighbour) pairs per launch, random rows of a 4 GB code table, 16 launches per layout and row stride "
      "(6 warm-up + 10 timed with HIP events).  Algorithmic bytes per evaluation = m + 8 (SURVEY 8(d)).\n")
print("## the tool's own lines (HIP events), un-profiled pass = the kernel-trace pass\n")
for l in open(os.path.join(out, "k2_trace.jsonl")):
    if l.startswith("{"):
        d = json.loads(l)
        print(f"* m = {d['m']}, rows {d['code_stride']} B apart: {d['avg_launch_us']} us per launch (min {d['min_launch_us']}), {d['achieved']} GB/s algorithmic = {d['frac']:.3f} of 8 TB/s, "
              f"{d['rows_per_s']} G rows/s, table {d['code_table_bytes']/1e9:.1f} GB")
print()
f = glob.glob(os.path.join(out, "trace/**/*kernel_trace.csv"), recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "pqdist_stream_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print("## rocprofv3 --kernel-trace: launches of pqdist_stream_kernel, in order (16 per layout: m = 32, 70, 70 at stride 128, 74, 74 at stride 128)\n")
    print("| layout | launches | avg us (last 10) | min us | max us | kernel |")
    print("|---|---|---|---|---|---|")
    trace = {}
    for i, m in enumerate(LAYOUTS):
        grp = rows[i * NL:(i + 1) * NL][WARM:]
        if not grp:
            continue
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp]
        print(f"| m = {m} | {len(grp)} | {sum(du)/len(du):.1f} | {min(du):.1f} | {max(du):.1f} | {grp[0]['Kernel_Name'].split('(')[0][:60]} |")
        key = ("m%s_stride128" % m.split()[0]) if "128" in m else ("m%s_packed" % m)
        trace[key] = {"avg_us": round(sum(du) / len(du), 1), "min_us": round(min(du), 1), "launches": len(grp)}
    print()
    # (what bench.py quotes beside its HIP-event time: copy to profiles/k2_alone_trace.json)
    json.dump({"source": "rocprofv3 --kernel-trace of tools/k2_alone.py --big (tools/profile_k2.sh)", "layouts": trace}, open(os.path.join(os.path.dirname(out), "profiles_out", "k2_alone_trace.json"), "w"))
for name, label in (("pmc_fetch", "FETCH_SIZE (KB; tallies every read request at 64 bytes although each is a 128-byte line: HBM bytes = 2 x FETCH_SIZE = TCC_EA0_RDREQ x 128, profiles/r04_traffic_calibration.md)"),
                    ("pmc_l2", "L2 / fabric counters"),
                    ("pmc_sq1", "where the wave cycles go (SQ, quad-cycles summed over waves)"),
                    ("pmc_sq2", "instructions and LDS bank conflicts (SQ)")):
    f = glob.glob(os.path.join(out, name + "/**/*counter_collection.csv"), recursive=True)
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f[0])) if "pqdist_stream_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    print(f"## {label}, per launch (own pass)\n")
    print("| layout | counter | avg per launch |")
    print("|---|---|---|")
    for i, m in enumerate(LAYOUTS):
        keep = set(ids[i * NL:(i + 1) * NL][WARM:])
        agg = {}
        for r in rows:
            if int(r["Dispatch_Id"]) in keep:
                agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for c, v in sorted(agg.items()):
            print(f"| m = {m} | {c} | {v/max(1,len(keep)):,.0f} |")
    print()
PY
cat gpurun_out/profiles_out/${TAG}_k2_alone.md | head -60
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_l2" "$OUT/pmc_sq1" "$OUT/pmc_sq2"
