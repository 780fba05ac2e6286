import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bang-billion-scale-ann_amd"))
import torch, bench
class Ctx: pass
ctx = Ctx(); ctx.dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for rep in range(2):
    r = bench.k2_alone(128, 32, "uint8", ctx, reps=10)
    print(os.environ.get("BANG_AMD_LIB", "default")[-30:], r["avg_launch_us"], r["rows_per_s"], r["frac"], flush=True)
