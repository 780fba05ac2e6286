"""bench.py leg: the REQUEST ceiling -- how many random requests past L2 per second this MI355X completes in the mix the search kernel issues
them (tools/request_ceiling.hip), measured live, beside the rate the search launch itself reached (rocprofv3 --pmc passes of the same run).

Why: per distance evaluation the search path sends ~3.8 requests past L2 that nothing ever re-uses (a 128-byte line for the code row, a line for
a filter word it must read, 1.8 scattered 4-byte filter stores; reference stages bang_search.cu:1140-1165, :1201-1241).  The byte roofline the
contract prescribes (`roofline.frac`: (m + 8) bytes per evaluation against 8 TB/s) cannot see that stream; this leg states the bound that
actually holds: requests per second."""
import ctypes as C
import os
import subprocess

from .common import ROOT

_SO = os.path.join(ROOT, "tools", "_build", "librequest_ceiling.so")
_SRC = os.path.join(ROOT, "tools", "request_ceiling.hip")


def build():
    """hipcc cross-compiles without a GPU; a profiled child (BANG_NO_BUILD) never compiles."""
    if os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    if os.environ.get("BANG_NO_BUILD"):
        raise RuntimeError(f"{_SO} is missing or stale and BANG_NO_BUILD is set")
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", _SO, _SRC])
    return _SO


MIXES = {"probes_only": 0, "search_mix": 1, "code_lines_only": 2, "probe_plus_store": 3}


def request_ceiling(filter_bytes=154 << 20, code_bytes=16 << 30, trips=256, waves_per_cu=16, mixes=("search_mix", "probes_only")):
    """-> {mix: G requests/s}.  filter_bytes: the visited filters of the queries resident at a time (12 per CU x 256 CUs x 50 KB = 154 MB);
    code_bytes: a table far beyond the 256 MB Infinity Cache."""
    lib = C.CDLL(build())
    lib.request_ceiling.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    out = {}
    for name in mixes:
        g, ms = C.c_double(), C.c_double()
        rc = lib.request_ceiling(MIXES[name], code_bytes, filter_bytes, trips, waves_per_cu, C.byref(g), C.byref(ms))
        out[name] = round(g.value, 2) if rc == 0 else None
    return out


def request_roofline(live, avg_launch_us, ceil):
    """The search launch's requests past L2 per second (from the live PMC passes: read requests = TCC_EA0_RDREQ_DRAM_32B x 32 / 128 -- every
    read request is a 128-byte line, profiles/r04_traffic_calibration.md -- + TCC_EA0_WRREQ + the PCIe reads) against the measured ceiling."""
    if not live or not live.get("bytes") or not avg_launch_us:
        return None
    rd = live["hbm_read"] / 128.0 + live.get("pcie_read", 0) / 128.0
    wr = float(live.get("write_requests", 0))
    per_launch = rd + wr
    per_s = per_launch / (avg_launch_us * 1e-6) / 1e9
    c = (ceil or {}).get("search_mix")
    return {"bound": "requests past L2 (random, never re-used)", "requests_per_launch": int(per_launch), "read_requests": int(rd), "write_requests": int(wr),
            "achieved": round(per_s, 2), "peak": c, "unit": "G requests/s", "frac": (round(per_s / c, 4) if c else None),
            "peak_is": "tools/request_ceiling.hip measured in THIS run: per lane and trip one random 128-byte code line (16 GB table), one 4-byte probe past L1 "
                       "and two 4-byte stores into a 154 MB table (the resident queries' filters), 4 trips in flight, 16 waves per CU",
            "ceiling_probes_only": (ceil or {}).get("probes_only")}
