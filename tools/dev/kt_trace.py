"""Per-launch front-kernel times of lane 0 from a BANG_KT_TRACE dump (launch-per-iteration mode)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
n, wgs = int(raw[0]), int(raw[1])
kt = raw[2:].reshape(n, wgs, 2).astype(np.float64) / 100.0
for it in list(range(0, min(n, 8))) + [10, 20, 40, 60, 70, 75, 80, 90, 100, 110]:
    if it >= n: break
    u = kt[it, :, 0] > 0
    if not u.any(): continue
    span = kt[it, u, 1].max() - kt[it, u, 0].min()
    per = (kt[it, u, 1] - kt[it, u, 0])
    print(f"launch {it:3d}: WGs {int(u.sum()):3d} span {span:7.2f} us, per-WG mean {per.mean():7.2f} max {per.max():7.2f}")
