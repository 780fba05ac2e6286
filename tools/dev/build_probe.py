#!/usr/bin/env python3
"""Builds a structured index of N points on the GPU and prints the time and the peak HBM use of every stage's end: how large can the recall-verified leg be?
    python tools/dev/build_probe.py N"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
import torch  # noqa: E402
from bang_amd import index_build  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000


def log(*a):
    print(*a, f"[HBM now {torch.cuda.memory_allocated() / 2**30:.0f} GiB, peak {torch.cuda.max_memory_allocated() / 2**30:.0f} GiB]", flush=True)


try:
    t0 = time.time()
    ix, q, gi, gd = index_build.make_index_large(N, 128, "uint8", 64, 70, 10000, K=10, n_clusters=N // 10000, device="cuda", log=log, select="groupmin", probes=12)
    print("OK", round(time.time() - t0, 1), "s; peak", round(torch.cuda.max_memory_allocated() / 2**30, 1), "GiB; graph", ix.graph.nbytes / 1e9, "GB")
except Exception:
    traceback.print_exc()
    print("peak GiB", torch.cuda.max_memory_allocated() / 2**30)
