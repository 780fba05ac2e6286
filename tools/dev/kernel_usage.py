#!/usr/bin/env python3
"""VGPRs / scratch / LDS of every kernel instance: parses `hipcc -Rpass-analysis=kernel-resource-usage` remarks.
    hipcc ... -Rpass-analysis=kernel-resource-usage -c csrc/bang_search.hip -o /tmp/x.o 2> usage.txt ; kernel_usage.py usage.txt [filter]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].strip()
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt in dn:
        sc, oc = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
        print(f"{dn[:90]:90s} vgpr {g('VGPRs'):3d} agpr {g('AGPRs'):3d} scratch {sc:4d} sgpr {g('SGPRs'):3d} occ {oc}")
