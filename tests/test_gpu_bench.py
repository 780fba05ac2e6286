"""bench.py's own measurement of roofline.traffic: two child runs of the primary configuration under rocprofv3 --pmc (pass A: FETCH_SIZE +
TCC_EA0_RDREQ_DRAM_32B, pass B: WRITE_SIZE + TCC_EA0_RDREQ_IO_32B + TCC_EA0_WRREQ), before the parent touches the GPU
(MI355X_MICROARCH.md: HBM bytes, separate passes; the byte-exact counters are calibrated in profiles/r04_traffic_calibration.md), and the
order of the line's keys (the driver's record keeps the leading scalars)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500, method="thread")
def test_bench_measures_hbm_traffic_in_the_same_run(libbang):
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not installed")
    cmd = [sys.executable, "bench.py", "--workload", "small", "--L", "46", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--legs", "none"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    rf = d["roofline"]
    if "live rocprofv3 --pmc" not in str(rf.get("traffic_note")):   # (a cold box: the profiled child had to page the whole stack in and ran out of its time; once more, warm)
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        rf = d["roofline"]
    assert "live rocprofv3 --pmc" in str(rf.get("traffic_note")), (rf.get("traffic_note"), r.stderr[-1500:])
    # every evaluation reads its code row and writes a distance: the HBM-side bytes cannot be fewer than ... well, they can be served by
    # L2 on a 100 K-point index; what must hold is that the counters saw the launches (> 0) and stay within a sane multiple
    assert 0 < rf["traffic"] < 200 * rf["algorithmic_bytes_per_launch"]
    assert d["config"]["parity_vs_oracle_first_64"] is True
    by = rf["traffic_by_stream"]
    assert set(by) >= {"code_rows", "filter_reads", "filter_writes"} and by["code_rows"] > 0 and by["filter_writes"] > 0
    assert rf["traffic_over_algorithmic"] == pytest.approx(rf["traffic"] / rf["algorithmic_bytes_per_launch"], rel=1e-3) and rf["hbm_traffic_GBps"] > 0
    # scalars first: the keys the record must keep sit in front of the prose and the nested objects
    first = list(d["config"])[:32]
    for key in ("recall_gated_qps", "recall_gated_N", "recall_gated_recall", "workload", "L", "recall_at_10", "k2_alone_frac", "traffic_over_algorithmic",
                "qps_incl_init", "shard_ms_2500", "shard_ms_1250", "projected_speedup_4", "gather_ms_world1", "walker_qps", "rccl_world_seen", "sift300m_qps", "sift1m_qps"):
        assert key in first, key
    assert list(d["config"])[:3] == ["recall_gated_qps", "recall_gated_N", "recall_gated_recall"]
    rkeys = list(rf)
    assert rkeys.index("traffic") < rkeys.index("kernel") and rkeys.index("k2_alone_frac") < rkeys.index("kernel")


def test_bench_without_live_traffic_quotes_the_committed_passes(libbang):
    cmd = [sys.executable, "bench.py", "--workload", "sift1m", "--graph", "device", "--L", "70", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--legs", "none", "--no-live-traffic"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "NOT re-measured" in d["roofline"]["traffic_note"]


def test_bench_shards_and_walker_legs_on_the_primary_engine(libbang):
    """The legs that run on the primary engine (one load): the shard sweep with the world-size-1 RCCL gather and the projected speed-ups
    WITH the gather, and the north-star data flow (C++ walker threads reading the pull rows) -- here on a reduced SIFT1B-shape index."""
    cmd = [sys.executable, "bench.py", "--workload", "sift1b_shape", "--shape-n", "4000000", "--queries", "2048", "--L", "40", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--legs", "shards,walker", "--no-live-traffic"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert "shape-only" in d["metric"].lower() and d["vs_baseline"] is None
    sh = c["at_shards"]
    for q in (1024, 512, 256):
        assert c[f"shard_ms_{q}"] > 0 and sh[f"shard_{q}"]["result_properties_ok"] is True, sh
    assert c["gather_ms_world1"] is not None and 0 < c["gather_ms_world1"] < 5.0, sh
    for W, q in ((2, 1024), (4, 512), (8, 256)):
        assert c[f"projected_speedup_{W}"] == pytest.approx(d["ms_per_step"] / (c[f"shard_ms_{q}"] + c["gather_ms_world1"]), rel=2e-3)
    w = c["at_sift1b_shape_walker"]
    assert c["walker_qps"] > 0 and c["walker_N"] == 4000000 and w["ids_equal_pulled_run"] is True and w["result_properties_ok"] is True, w
    assert "walker threads read the 256-byte adjacency rows" in w["host_loop"]
