"""Summarise a BANG_PW_TRACE dump of the persistent search kernel: per-workgroup phase times (100 MHz ticks -> us)."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
iters, wgs = int(raw[0]), int(raw[1])
kt = raw[2:].reshape(iters, wgs, 4).astype(np.float64) / 100.0      # us
used = kt[:, :, 0] > 0
nit = used.sum(axis=0)
live = nit > 0
print(f"iterations stored {iters}, workgroups {int(live.sum())}, iterations per WG min/mean/max {nit[live].min()}/{nit[live].mean():.1f}/{nit[live].max()}")
t0 = kt[:, :, 0][used].min()
first = np.where(used, kt[:, :, 0], np.inf).min(axis=0)[live] - t0
last = kt[:, :, 3].max(axis=0)[live] - t0
print(f"first go-seen per WG: min {first.min():.1f} max {first.max():.1f} us; last stamp per WG: min {last.min():.1f} mean {last.mean():.1f} max {last.max():.1f} us")
front = (kt[:, :, 1] - kt[:, :, 0])[used]
flag = (kt[:, :, 2] - kt[:, :, 1])[used]
back = (kt[:, :, 3] - kt[:, :, 2])[used & (kt[:, :, 3] > 0)]
wait = (kt[1:, :, 0] - kt[:-1, :, 3])[used[1:] & used[:-1] & (kt[:-1, :, 3] > 0)]
for name, v in (("front", front), ("flag", flag), ("back", back), ("wait-for-go", wait)):
    print(f"{name:12s} n={v.size:7d} mean {v.mean():7.2f} us  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  p99 {np.percentile(v, 99):7.2f}  sum/WG {v.sum() / live.sum() / 1000:.3f} ms")
# front time by iteration number (first iterations carry all queries of the block)
for it in (1, 2, 5, 20, 50, 70, 80, 100):
    if it < iters and used[it].any():
        f = (kt[it, :, 1] - kt[it, :, 0])[used[it]]
        w = (kt[it, :, 0] - kt[it - 1, :, 3])[used[it] & used[it - 1]] if it > 1 else np.zeros(1)
        print(f"  iter {it:3d}: WGs {int(used[it].sum()):3d} front mean {f.mean():6.2f} us, wait mean {w.mean():6.2f} us")
