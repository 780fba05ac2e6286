import json, os, sys
sys.path.insert(0, os.getcwd())
from tools.bench_legs import ceiling
for waves in (1, 2, 3, 4, 6, 8, 12, 16):
    r = ceiling.request_ceiling(filter_bytes=16 << 20, waves_per_cu=waves, mixes=("code_lines_only",))
    print(json.dumps(dict(waves_per_cu=waves, lines_in_flight_per_cu=waves * 64 * 4, **r)), flush=True)
