#!/usr/bin/env python3
"""Joins the per-launch counters of tools/calibrate_traffic.sh with the byte counts tools/traffic_calib prints: what each gfx950
memory-side counter reports per known byte, for the access shapes of the search path."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
known = json.loads(open(os.path.join(out, "known.json")).read().strip().splitlines()[-1])
order = known["launch_order"]
vals = defaultdict(dict)          # kernel -> counter -> value of the SECOND launch
for p in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(p)) if any(k in r["Kernel_Name"] for k in ("k_rows", "k_probe4", "k_store4", "k_stream"))]   # (not the memset's fill kernel)
    disp = sorted({int(r["Dispatch_Id"]) for r in rows})
    # launches come in two rounds of len(order): the second round is the warm one
    second = disp[len(order):2 * len(order)] if len(disp) >= 2 * len(order) else disp[-len(order):]
    extra = disp[2 * len(order):][-2:]                        # (the uncached-memory probes / stores: last round)
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d in extra:
            k = ("probe4_uncached", "store4_uncached")[extra.index(d)]
            vals[k][r["Counter_Name"]] = vals[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if d in second:
            vals[order[second.index(d)]][r["Counter_Name"]] = vals[order[second.index(d)]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("# gfx950 memory-side counters against known byte counts (tools/traffic_calib.hip, 16 GB table, second launch of each kernel)\n")
print("FETCH_SIZE / WRITE_SIZE are in KB; `*_32B` counters count 32-byte units (a 64-byte request counts 2, a 128-byte request 4).\n")
names = sorted({c for k in vals.values() for c in k})
print("| kernel | known | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k in order:
    kn = known[k]
    print(f"| {k} | {json.dumps(kn)} | " + " | ".join(f"{vals[k].get(c, float('nan')):.4g}" for c in names) + " |")
print()


def g(k, c):
    return vals[k].get(c, float("nan"))


print("## bytes per known unit\n")
print("| kernel | unit | FETCH_SIZE x 1024 | + 64 x RDREQ_128B | 32 x (RDREQ_DRAM_32B + IO_32B) | 32 n32 + 64 n64 + 128 n128 | WRITE_SIZE x 1024 | 32 x WRREQ_WRITE_DRAM_32B | requests (RDREQ / WRREQ) |")
print("|---|---|---|---|---|---|---|---|---|")
for k, unit, n in (("rows128", "row of 128 B", known["rows128"]["rows"]), ("rows70", "row of 70 B (packed)", known["rows70"]["rows"]),
                   ("probe4", "4-byte probe", known["probe4"]["probes"]), ("store4", "4-byte store", known["store4"]["stores"]),
                   ("stream", "byte streamed", known["stream"]["bytes"]),
                   ("probe4_uncached", "4-byte probe, uncached device memory", known["probe4"]["probes"]),
                   ("store4_uncached", "4-byte store, uncached device memory", known["store4"]["stores"])):
    if k not in vals:
        continue
    f = g(k, "FETCH_SIZE") * 1024
    f2 = f + 64 * g(k, "TCC_EA0_RDREQ_128B_sum")
    dr = 32 * (g(k, "TCC_EA0_RDREQ_DRAM_32B_sum") + g(k, "TCC_EA0_RDREQ_IO_32B_sum"))
    sz = 32 * g(k, "TCC_EA0_RDREQ_32B_sum") + 64 * g(k, "TCC_EA0_RDREQ_64B_sum") + 128 * g(k, "TCC_EA0_RDREQ_128B_sum")
    w = g(k, "WRITE_SIZE") * 1024
    wd = 32 * g(k, "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum")
    print(f"| {k} | {unit} | {f / n:.3f} | {f2 / n:.3f} | {dr / n:.3f} | {sz / n:.3f} | {w / n:.3f} | {wd / n:.3f} | "
          f"{g(k, 'TCC_EA0_RDREQ_sum') / n:.3f} / {g(k, 'TCC_EA0_WRREQ_sum') / n:.3f} |")
