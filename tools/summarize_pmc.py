#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of tools/profile2.sh into a markdown summary: per-kernel times of this repo's kernels, the duration
of the timed launches of the search kernel, and every PMC counter per launch of that kernel (each from its own pass), plus the
small JSON with the HBM traffic per launch that bench.py quotes as roofline.traffic.

Units / corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KB (TCC_EA0_RDREQ/WRREQ based).  FETCH_SIZE
tallies every read request of gfx950's L2 at 64 bytes although each is a 128-byte line -- for EVERY access shape of this path
(4-byte probes, 128-byte code rows, streamed reads alike: tools/traffic_calib.hip on known byte counts,
profiles/r04_traffic_calibration.md) -- so the read bytes are 2 x FETCH_SIZE, and TCC_EA0_RDREQ_DRAM_32B x 32 (pass `dram`) gives them
byte-exactly; WRITE_SIZE is exact (32 bytes per scattered 4-byte store).  `traffic` = DRAM reads + writes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else None
OURS = ("search_kernel", "front_kernel", "back_kernel", "rerank_kernel", "center_queries_kernel", "lut_build_kernel", "init_state_kernel")


def find(pattern):
    r = glob.glob(os.path.join(out, pattern), recursive=True)
    return r[0] if r else None


def short(name):
    base = name.split("(")[0]
    for k in OURS:
        if k in base:
            return k + (base[base.index("<"):] if "<" in base and k in ("front_kernel", "search_kernel") else "")
    return None


bench = {}
print(f"# rocprofv3 summary: {os.path.basename(out)}\n")
for p in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    t = os.path.basename(p)[6:-5]
    for line in open(p):
        if line.startswith("{"):
            bench[t] = json.loads(line)
if "trace" in bench:
    d = bench["trace"]
    rf = d.get("roofline") or {}
    print(f"bench line under the kernel trace: value={d['value']} {d['unit']}, ms_per_step={d['ms_per_step']}, L={d['config']['L']}, "
          f"graph={d['config']['graph']}, iterations={d['config']['iterations']}, workload={d['config']['workload'][:70]}...\n")
    print(f"its roofline object: achieved={rf.get('achieved')} GB/s, frac={rf.get('frac')}, avg_launch_us={rf.get('avg_launch_us')}, "
          f"algorithmic_bytes_per_launch={rf.get('algorithmic_bytes_per_launch')}\n")
print("(bench lines under the PMC passes are slowed down by counter collection: not performance numbers)\n")
steps = (bench.get("trace") or {}).get("steps") or 3

stats = find("trace/**/*kernel_stats.csv")
if stats:
    print("## kernel stats (rocprofv3 --kernel-trace --stats), this repo's kernels\n")
    print("| kernel | calls | total ms | avg us | min us | max us |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(stats)):
        s = short(r["Name"])
        if s:
            print(f"| {s} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                  f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
    print()


def dominant(rows, key):
    """rows of the dominant kernel (search_kernel if the run used it, else the persistent/front kernel), timed launches only."""
    for name in ("search_kernel", "front_kernel"):
        sel = [r for r in rows if name in r[key]]
        if sel:
            return name, sel
    return None, []


timed_avg_us = None
dom_name = None
ktrace = find("trace/**/*kernel_trace.csv")
if ktrace:
    rows = list(csv.DictReader(open(ktrace)))
    dom_name, sel = dominant(rows, "Kernel_Name")
    if sel:
        sel.sort(key=lambda r: int(r["Start_Timestamp"]))
        one_per_batch = len(sel) < 400                       # one launch per batch vs a launch per iteration
        last = sel[-steps:] if one_per_batch else sel
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
        timed_avg_us = sum(durs) / len(durs)
        what = f"the {len(last)} timed launches only" if one_per_batch else f"all {len(last)} launches"
        print(f"**{dom_name}**, {what} (kernel trace): avg {timed_avg_us:.1f} us (min {min(durs):.1f}, max {max(durs):.1f}); "
              f"bench.py's in-kernel timer for the timed launches: {((bench.get('trace') or {}).get('roofline') or {}).get('avg_launch_us')} us\n")

counters = {}
print("## PMC counters per launch of the dominant kernel (each group from its own pass)\n")
print("| pass | counter | avg per launch | launches |")
print("|---|---|---|---|")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    f = find(os.path.basename(d) + "/**/*counter_collection.csv")
    if not f:
        continue
    rows = list(csv.DictReader(open(f)))
    name, sel = dominant(rows, "Kernel_Name")
    if not sel:
        continue
    ids = sorted({int(r["Dispatch_Id"]) for r in sel})
    nsteps = (bench.get(os.path.basename(d)) or {}).get("steps") or steps
    keep = set(ids[-nsteps:]) if len(ids) < 400 else set(ids)
    agg = defaultdict(float)
    for r in sel:
        if int(r["Dispatch_Id"]) in keep:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for c, v in sorted(agg.items()):
        counters[c] = v / len(keep)
        print(f"| {os.path.basename(d)} | {c} | {v / len(keep):,.0f} | {len(keep)} |")
print()

c = counters
print("## derived\n")
if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
    fetch_raw, write = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    dram = c.get("TCC_EA0_RDREQ_DRAM_32B_sum", 0.0) * 32
    io = c.get("TCC_EA0_RDREQ_IO_32B_sum", 0.0) * 32
    fetch = dram if dram else 2 * fetch_raw - io                # (every request is a 128-byte line tallied at 64; the PCIe share is not HBM traffic)
    print(f"* HBM traffic per launch: reads {fetch/1e9:.3f} GB (" + (f"TCC_EA0_RDREQ_DRAM_32B x 32; " if dram else "2 x FETCH_SIZE - PCIe reads; ") +
          f"FETCH_SIZE raw {fetch_raw/1e9:.3f} GB tallies each 128-byte request at 64) + writes WRITE_SIZE {write/1e9:.3f} GB = {(fetch+write)/1e9:.3f} GB"
          + (f"; {io/1e9:.3f} GB more read over PCIe (TCC_EA0_RDREQ_IO_32B x 32: pulled adjacency rows)" if io else ""))
    rf = (bench.get("trace") or {}).get("roofline") or {}
    alg = rf.get("algorithmic_bytes_per_launch")
    if alg:
        print(f"* algorithmic bytes per launch {alg/1e9:.3f} GB -> traffic / algorithmic = {(fetch+write)/alg:.2f}x")
    if timed_avg_us:
        print(f"* over the {timed_avg_us:.0f} us of a launch: {(fetch+write)/timed_avg_us/1e3:.0f} GB/s of HBM-side traffic")
    if tag:
        js = {"search_kernel_hbm_bytes_per_launch": round(fetch + write), "kernel": dom_name,
              "fetch_size_raw_bytes_per_launch": round(fetch_raw), "hbm_read_bytes_per_launch": round(fetch), "pcie_read_bytes_per_launch": round(io),
              "write_bytes_per_launch": round(write), "algorithmic_bytes_per_launch": alg,
              "rocprof_timed_launches_avg_us": timed_avg_us, "bench_in_kernel_timer_avg_us": rf.get("avg_launch_us"),
              "note": "reads = TCC_EA0_RDREQ_DRAM_32B x 32 where that pass ran, else 2 x FETCH_SIZE (every request a 128-byte line tallied at 64: "
                      "profiles/r04_traffic_calibration.md) less the PCIe reads; writes = WRITE_SIZE; separate --pmc passes of the same bench command"}
        json.dump(js, open(os.path.join(out, f"traffic_{tag}.json"), "w"), indent=1)
if "TCC_HIT_sum" in c:
    h, m = c["TCC_HIT_sum"], c.get("TCC_MISS_sum", 0.0)
    print(f"* L2: {h:,.0f} hits, {m:,.0f} misses per launch -> hit rate {h/(h+m+1e-9):.3f}; {c.get('TCC_REQ_sum',0):,.0f} requests, "
          f"{c.get('TCC_ATOMIC_sum',0):,.0f} atomics")
if "TCC_EA0_RDREQ_sum" in c:
    print(f"* fabric (EA) per launch: {c['TCC_EA0_RDREQ_sum']:,.0f} read requests ({c.get('TCC_EA0_RDREQ_32B_sum',0):,.0f} of them 32 B), "
          f"{c.get('TCC_EA0_WRREQ_sum',0):,.0f} write requests, {c.get('TCC_EA0_ATOMIC_sum',0):,.0f} atomics")
    if timed_avg_us:
        print(f"* = {(c['TCC_EA0_RDREQ_sum'] + c.get('TCC_EA0_WRREQ_sum',0))/timed_avg_us/1e3:.1f} G fabric requests/s")
if "TCP_TCC_READ_REQ_sum" in c:
    print(f"* L1 -> L2 per launch: {c['TCP_TCC_READ_REQ_sum']:,.0f} reads, {c.get('TCP_TCC_WRITE_REQ_sum',0):,.0f} writes, "
          f"{c.get('TCP_TCC_ATOMIC_WITH_RET_REQ_sum',0) + c.get('TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum',0):,.0f} atomics")
if "SQ_WAVE_CYCLES" in c:
    wc = c["SQ_WAVE_CYCLES"]
    print(f"* wave-cycles (quad-cycles, summed over waves): {wc:,.0f}; waiting (s_waitcnt / barrier) {c.get('SQ_WAIT_ANY',0)/wc:.1%}, "
          f"issue-stalled {c.get('SQ_WAIT_INST_ANY',0)/wc:.1%}; active: VALU {c.get('SQ_ACTIVE_INST_VALU',0)/wc:.1%}, "
          f"LDS {c.get('SQ_ACTIVE_INST_LDS',0)/wc:.1%}, VMEM {c.get('SQ_ACTIVE_INST_VMEM',0)/wc:.1%}, scalar {c.get('SQ_ACTIVE_INST_SCA',0)/wc:.1%}")
if "SQ_INSTS_LDS" in c:
    print(f"* instructions per launch: VALU {c.get('SQ_INSTS_VALU',0):,.0f}, SALU {c.get('SQ_INSTS_SALU',0):,.0f}, LDS {c['SQ_INSTS_LDS']:,.0f}, "
          f"VMEM read {c.get('SQ_INSTS_VMEM_RD',0):,.0f}, VMEM write {c.get('SQ_INSTS_VMEM_WR',0):,.0f}; LDS bank-conflict cycles "
          f"{c.get('SQ_LDS_BANK_CONFLICT',0):,.0f} of {c.get('SQ_LDS_IDX_ACTIVE',0):,.0f} LDS-active cycles")
