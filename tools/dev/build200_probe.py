import sys, os, time, traceback
sys.path.insert(0, "bang-billion-scale-ann_amd")
import torch
from bang_amd import index_build
log = lambda *a: print(*a, flush=True)
try:
    t0 = time.time()
    ix, q, gi, gd = index_build.make_index_large(200_000_000, 128, "uint8", 64, 70, 10000, K=10, n_clusters=20000, device="cuda", log=log, select="groupmin", probes=12)
    print("OK", time.time() - t0, torch.cuda.max_memory_allocated() / 2**30)
except Exception:
    traceback.print_exc()
    print("max alloc GiB", torch.cuda.max_memory_allocated() / 2**30)
