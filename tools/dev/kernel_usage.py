#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / occupancy of every kernel instance of a .hip file, from `hipcc -Rpass-analysis=kernel-resource-usage`.

    python tools/dev/kernel_usage.py [--file csrc/bang_search.hip] [--filter search_kernel] [--md profiles/r05_kernel_usage.md] [-D...]

Run after any kernel change: the long-row instances of the search kernel sit within a few registers of their 168-VGPR / 106-SGPR budget,
and one more live value becomes scratch traffic inside the row reduce."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "bang-billion-scale-ann_amd")


def usage_of(src, extra):
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-Rpass-analysis=kernel-resource-usage", "-c", src,
               "-o", os.path.join(td, "x.o")] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr[-4000:])
            raise SystemExit(r.returncode)
        return r.stderr


def rows(text, flt):
    out = []
    for b in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = b.split("\n")[0].split(" [-R")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"^void ", "", dn).replace("(SearchArgs)", "").replace("(bang_iter_params, unsigned int)", "")
        if flt in dn:
            out.append(dict(kernel=dn, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("SGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                            occ=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", default=os.path.join(PKG, "csrc", "bang_search.hip"))
    ap.add_argument("--filter", default="")
    ap.add_argument("--md", default=None, help="also write a markdown table here")
    a, extra = ap.parse_known_args()
    if os.path.basename(a.file) == "bang_search.hip" and not any(x.startswith("-DBANG_SEARCH_PART") for x in extra):
        # the search kernel's instances live in two translation units of the one file (Makefile: part 0 under the ILP scheduling strategy, part 1 under the default)
        sched = [] if any("sched-strategy" in x for x in extra) or os.environ.get("BANG_USAGE_NO_SCHED") else ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
        rs = rows(usage_of(a.file, extra + sched), a.filter) + rows(usage_of(a.file, extra + ["-DBANG_SEARCH_PART=1"]), a.filter)
    else:
        rs = rows(usage_of(a.file, extra), a.filter)
    for r in rs:
        print(f"{r['kernel'][:70]:70s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} scratch {r['scratch']:4d} occ {r['occ']}")
    if a.md:
        with open(a.md, "w") as f:
            f.write(f"# Register budget per kernel instance: `{os.path.relpath(a.file, ROOT)}`\n\n")
            f.write("`python tools/dev/kernel_usage.py --md <this file>` (hipcc `-Rpass-analysis=kernel-resource-usage`, gfx950).  Template arguments of "
                    "`search_kernel`: `<PSZ, NDW, ALIGNED, NHI, HOST, SPEC>` -- `<2, 18, true, 58, false, *>` is the SIFT1B layout (m = 70, rows 128 B apart), self-paced "
                    "(SPEC = code rows requested with the filter probes); `<2, 19, *, 22, false, *>` DEEP100M (m = 74); `<4, 8, true, 0, false, false>` SIFT1M (m = 32).  Budget: 168 VGPRs for the 12-wave (768-thread) "
                    "instances, 128 for the 16-wave ones; 106 SGPRs (the compiler always reports the cap).  Dynamic LDS is sized at launch.  The instances are compiled in two translation units "
                    "(Makefile: `bang_search.o` under `-amdgpu-sched-strategy=iterative-ilp`, `bang_search_b.o` -- 96+ chunks, unaligned 74-chunk rows -- under the default scheduler).\n\n")
            f.write("| kernel instance | VGPRs | AGPRs | SGPRs | scratch B/lane | waves/SIMD |\n|---|---|---|---|---|---|\n")
            for r in rs:
                f.write(f"| `{r['kernel']}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | {r['scratch']} | {r['occ']} |\n")


if __name__ == "__main__":
    main()
