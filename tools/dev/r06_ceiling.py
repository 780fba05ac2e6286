#!/usr/bin/env python3
"""Sweep of tools/request_ceiling.hip: requests past L2 per second by mix, filter-table size and waves per CU (round 6)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.bench_legs import ceiling  # noqa: E402

rows = []
for filt_mb in (16, 154, 4096):
    for waves in (8, 12, 16):
        r = ceiling.request_ceiling(filter_bytes=filt_mb << 20, waves_per_cu=waves, mixes=("search_mix", "probes_only", "code_lines_only", "probe_plus_store"))
        rows.append(dict(filter_MB=filt_mb, waves_per_cu=waves, **r))
        print(json.dumps(rows[-1]), flush=True)
