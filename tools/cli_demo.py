#!/usr/bin/env python3
"""Writes the SIFT1M-like synthetic index in the reference's file formats and runs the drop-in `bang_search` CLI on it
(auto sweep, reference table format), for both graph placements.  Output goes to stdout (committed under profiles/)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))

import torch  # noqa: E402
from bang_amd import formats, synth  # noqa: E402

N, Q = int(os.environ.get("DEMO_N", "1000000")), 10000
ix, q, gt_i, gt_d = synth.make_index(N, 128, "uint8", 64, 32, Q, K=10, n_clusters=256,
                                     device="cuda" if torch.cuda.is_available() else "cpu")
d = tempfile.mkdtemp(prefix="bang_demo_")
prefix = os.path.join(d, "sift1m_like")
formats.write_index(prefix, ix)
formats.write_bin(prefix + "_query.bin", q)
formats.write_truthset(prefix + "_gt.bin", gt_i, gt_d)
exe = os.path.join(ROOT, "bang-billion-scale-ann_amd", "bin", "bang_search")
for graph in ("host", "device"):
    cmd = [exe, prefix, prefix + "_query.bin", prefix + "_gt.bin", str(Q), "10", "uint8", "l2", "auto"]
    print(f"$ BANG_GRAPH={graph} bang_search <prefix> <query.bin> <gt.bin> {Q} 10 uint8 l2 auto   # N={N}", flush=True)
    env = dict(os.environ, BANG_GRAPH=graph)
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    lines = out.stdout.splitlines()
    keep = [l for l in lines if not l[:1].isdigit() or int(l.split("\t")[0]) <= 130]    # table up to L = 130
    print("\n".join(keep), flush=True)
    if out.returncode != 0:
        print("rc", out.returncode, out.stderr[-500:])
