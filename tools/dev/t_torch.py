import sys, time
sys.path.insert(0, "bang-billion-scale-ann_amd"); sys.path.insert(0, ".")
t0=time.time()
import torch
print("torch import", time.time()-t0, torch.cuda.is_available(), torch.version.hip)
x = torch.randn(1000,1000,device="cuda"); print((x@x).sum().item())
import numpy as np
import bang_amd
from bang_amd import synth
from oracle import oracle as O
print("devices", bang_amd.device_count())
t0=time.time()
ix,q,gi,gd = synth.make_index(20000,128,"uint8",64,70,500,device="cuda")
print("build 20k on gpu", time.time()-t0)
orc = O.Oracle(ix)
ids_o, d_o = orc.search(q,10,64)
for graph in (0,1):
    with bang_amd.Engine("uint8", graph=graph, timing=1) as e:
        e.load_index(ix); e.set_searchparams(10,64); e.alloc(500); e.init(500)
        ids,d = e.query(q)
        torch.cuda.synchronize()
        print("graph",graph,"equal", np.array_equal(ids,ids_o), np.array_equal(d,d_o), e.stats(), "recall", O.recall(gi,gd,ids,10))
import subprocess
print(open("/proc/self/maps").read().count("libamdhip64"))
print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:2])
