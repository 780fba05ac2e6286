#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs of tools/profile.sh into a markdown summary (this repo's kernels only: time, calls, HBM bytes
per launch from the two PMC passes) and a small JSON with the per-launch HBM traffic of the front kernel that bench.py
reports as roofline.traffic.

rocprofv3's FETCH_SIZE / WRITE_SIZE are in KB (TCC_EA0_RDREQ/WRREQ based).  Guide correction (MI355X_MICROARCH.md, HBM):
FETCH_SIZE reads exactly half the bytes of a WIDE COALESCED read stream on gfx950 (128-B requests tallied at 64 B).  The
front kernel's reads are random 4-74-byte probes (64-B requests), which that correction does not apply to and which the
guide calls uncalibrated; both the raw value and the x2 upper bound are listed, traffic uses the raw value."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else None
OURS = ("front_kernel", "back_kernel", "rerank_kernel", "center_queries_kernel", "lut_build_kernel", "init_state_kernel")


def find(pattern):
    r = glob.glob(os.path.join(out, pattern), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.split("(")[0]
    for k in OURS:
        if k in name:
            return k + (name[name.index("<"):] if "<" in name and k == "front_kernel" else "")
    return None


print(f"# rocprofv3 summary: {out}\n")
bench = {}
for t in ("trace", "pmc_fetch", "pmc_write"):
    p = os.path.join(out, f"bench_{t}.json")
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                d = json.loads(line)
                bench[t] = d
                print(f"bench line under `{t}`: value={d['value']} {d['unit']}, ms_per_step={d['ms_per_step']}, "
                      f"L={d['config']['L']}, graph={d['config']['graph']}, workload={d['config']['workload'][:60]}..., "
                      f"roofline={d['roofline']}\n")
print("(the bench lines under the PMC passes are slowed down by counter collection: not performance numbers)\n")

stats = find("trace/**/*kernel_stats.csv")
front_avg_us = None
if stats:
    print("## kernel stats (rocprofv3 --kernel-trace --stats), this repo's kernels\n")
    print("| kernel | calls | total ms | avg us | min us | max us |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(stats)):
        s = short(r["Name"])
        if not s:
            continue
        if s.startswith("front_kernel"):
            front_avg_us = float(r["AverageNs"]) / 1e3
        print(f"| {s} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
              f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
    print()

# the timed steps alone: the last `steps` dispatches of the front/search kernel in the kernel trace (bench.py runs an L sweep
# and warm-up launches of the same kernel first, which the per-kernel average above also contains)
timed_avg_us = None
ktrace = find("trace/**/*kernel_trace.csv")
steps = ((bench.get("trace") or {}).get("steps")) or 0
if ktrace and steps:
    rows = [r for r in csv.DictReader(open(ktrace)) if "front_kernel" in r.get("Kernel_Name", "")]
    persistent_run = "PERSIST" in str(((bench.get("trace") or {}).get("roofline") or {}).get("kernel", ""))
    if rows and persistent_run:
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        last = rows[-steps:]
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
        timed_avg_us = sum(durs) / len(durs)
        print(f"search kernel, the {steps} timed launches only (kernel trace): avg {timed_avg_us:.1f} us "
              f"(min {min(durs):.1f}, max {max(durs):.1f}); bench.py's in-kernel timer for the same launches: "
              f"{((bench.get('trace') or {}).get('roofline') or {}).get('avg_launch_us')} us\n")

per = {}
for t, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(f"{t}/**/*counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    rows_c = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == counter and short(r["Kernel_Name"])]
    persistent_run = "PERSIST" in str(((bench.get(t) or {}).get("roofline") or {}).get("kernel", ""))
    nsteps = ((bench.get(t) or {}).get("steps")) or 0
    if persistent_run and nsteps:
        # one search launch per batch: keep only the timed steps' launches (the L sweep and the warm-up come first)
        fr = sorted((r for r in rows_c if short(r["Kernel_Name"]).startswith("front_kernel")), key=lambda r: int(r["Dispatch_Id"]))
        drop = {id(r) for r in fr[:-nsteps]}
        rows_c = [r for r in rows_c if id(r) not in drop]
    for r in rows_c:
        s = short(r["Kernel_Name"])
        a = agg[s]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    print(f"## {counter} per launch (separate `rocprofv3 --pmc {counter}` pass; KB as reported)\n")
    print("| kernel | launches | avg KB / launch | avg MB / launch |")
    print("|---|---|---|---|")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {n} | {v/n:.1f} | {v/n/1024:.3f} |")
        if k.startswith("front_kernel"):
            per[counter] = v / n * 1024.0
    print()

if tag and per:
    fetch, write = per.get("FETCH_SIZE", 0.0), per.get("WRITE_SIZE", 0.0)
    rf = (bench.get("trace") or {}).get("roofline") or {}
    persistent = "PERSIST" in str(rf.get("kernel", ""))
    js = {("search_kernel_hbm_bytes_per_launch" if persistent else "front_kernel_hbm_bytes_per_launch"): round(fetch + write),
          "host_loop": "persistent search kernel (one launch per batch)" if persistent else "launch per iteration",
          "fetch_bytes_per_launch_raw": round(fetch), "fetch_bytes_per_launch_x2_upper_bound": round(2 * fetch),
          "write_bytes_per_launch": round(write),
          "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"),
          "rocprof_front_kernel_avg_us": front_avg_us, "rocprof_timed_launches_avg_us": timed_avg_us,
          "bench_in_kernel_timer_avg_us": rf.get("avg_launch_us"),
          "note": "FETCH_SIZE raw (random 64-B requests: the gfx950 x2 correction for wide coalesced reads does not apply; "
                  "uncalibrated per the guide) + WRITE_SIZE, separate --pmc passes of the same bench command"}
    os.makedirs("profiles", exist_ok=True)
    json.dump(js, open(os.path.join("profiles", f"traffic_{tag}.json"), "w"), indent=1)
    print("## traffic JSON\n\n```\n" + json.dumps(js, indent=1) + "\n```")
