"""The drop-in flow through files (test_driver.cpp:338-557) at 5e6 points: an index built on the GPU, written in the reference's file
formats, loaded by `bang_load` from the files (streamed pread, graph in host RAM) and searched by BOTH harnesses -- our bin/bang_search
and the reference's unmodified test_driver.cpp on libbang.so -- whose printed recall column must be what the oracle computes for the
same files at the same L."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_file_flow_recall_column_equals_the_oracles(libbang, tmp_path):
    import bang_amd
    from bang_amd import formats, index_build
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import file_flow
    N, Q = 5_000_000, 2_000
    ix, q, gt_i, gt_d = index_build.make_index_large(N, 128, "uint8", 64, 70, Q, K=10, n_clusters=512, device="cuda")
    prefix = str(tmp_path / "flow")
    formats.write_index(prefix, ix)
    formats.write_bin(prefix + "_query.bin", q)
    formats.write_truthset(prefix + "_gt.bin", gt_i, gt_d)
    orc = O.Oracle(formats.read_index(prefix, "uint8", mmap_graph=True))            # the oracle reads the FILES too
    want = {}
    for L in (10, 22, 34, 46, 58, 70):
        ids_o, _ = orc.search(q, 10, L, nthreads=min(16, os.cpu_count() or 1))
        want[L] = O.recall(gt_i, gt_d, ids_o, 10)
    args = [prefix, prefix + "_query.bin", prefix + "_gt.bin", str(Q), "10", "uint8", "l2", "auto"]
    env = dict(os.environ, BANG_GRAPH="host")
    bins = {"ours": os.path.join(os.path.dirname(os.path.dirname(bang_amd.lib_path())), "bin", "bang_search"),
            "reference": os.path.join(ROOT, "oracle", "_ref", "ref_bang_search")}
    seen = 0
    for name, b in bins.items():
        if not os.path.exists(b):
            continue
        r = subprocess.run([b] + args, capture_output=True, text=True, env=env, timeout=1800)
        assert r.returncode == 0, (name, r.stdout[-1500:], r.stderr[-1500:])
        rows = file_flow.table(r.stdout)
        assert len(rows) >= 5 * 6
        for L, _ms, _qps, rec in rows:
            if L in want:
                assert abs(rec - want[L]) < 0.006, (name, L, rec, want[L])      # (the harness prints two decimals)
        assert max(rec for L, _a, _b, rec in rows if L <= 70) >= 90.0
        seen += 1
    assert seen >= 1


def test_multi_gpu_harness_on_the_c_abi(libbang, tmp_path, small_u8):
    """bin/bang_search_multi: the harness over several GPUs of a node, written on the C-ABI alone -- one PROCESS per GPU, one shared rows file,
    PEER ROWS (each rank's HBM slice exported / imported as a hipIpcMemHandle through a shared-memory mailbox), the shards' ids in one shared
    [Q][k] block.  Here both ranks run on GPU 0 (`share`).  The printed recall is the oracle's for the same files, rows did come from the
    sibling's allocation, and -- with the slices cut short -- from host memory too."""
    import bang_amd
    from bang_amd import formats
    from oracle import oracle as O
    ix, q, gt_i, gt_d = small_u8
    prefix = str(tmp_path / "multi")
    formats.write_index(prefix, ix)
    formats.write_bin(prefix + "_query.bin", q)
    formats.write_truthset(prefix + "_gt.bin", gt_i, gt_d)
    ids_o, _ = O.Oracle(ix).search(q, 10, 48)
    want = O.recall(gt_i, gt_d, ids_o, 10)
    exe = os.path.join(os.path.dirname(os.path.dirname(bang_amd.lib_path())), "bin", "bang_search_multi")
    args = [exe, prefix, prefix + "_query.bin", prefix + "_gt.bin", str(q.shape[0]), "10", "uint8", "48", "2", "share"]
    for slice_rows, host_expected in ((None, False), (str(ix.N // 3), True)):
        env = dict(os.environ)
        env.pop("BANG_PULL_ROWS_DIR", None)
        r = subprocess.run(args + ([f"slice={slice_rows}"] if slice_rows else []), capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
        line = [l for l in r.stdout.splitlines() if l and l[0].isdigit()][-1].split("\t")
        assert int(line[0]) == 2 and int(line[1]) == 48
        assert abs(float(line[4]) - want) < 0.006, (line, want)
        own, peer, host = (int(x) for x in line[5].split("/"))
        assert own > 0 and peer > 0 and (host > 0) == host_expected, line


def test_multi_gpu_harness_ends_when_a_rank_dies_without_a_word(libbang, tmp_path, small_u8):
    """a rank that dies abnormally (SIGKILL here, as the OOM killer or a GPU fault abort would end it) never sets the shared `failed`
    word itself.  The parent reaps in the order the ranks end and sets it, so the sibling leaves its barrier and the harness returns 1 within
    seconds instead of spinning until an outer time limit."""
    import time
    import bang_amd
    from bang_amd import formats
    ix, q, gt_i, gt_d = small_u8
    prefix = str(tmp_path / "multi")
    formats.write_index(prefix, ix)
    formats.write_bin(prefix + "_query.bin", q)
    formats.write_truthset(prefix + "_gt.bin", gt_i, gt_d)
    exe = os.path.join(os.path.dirname(os.path.dirname(bang_amd.lib_path())), "bin", "bang_search_multi")
    args = [exe, prefix, prefix + "_query.bin", prefix + "_gt.bin", str(q.shape[0]), "10", "uint8", "48", "2", "share"]
    for victim in ("1", "0"):
        env = dict(os.environ, BANG_MULTI_TEST_KILL_RANK=victim)
        env.pop("BANG_PULL_ROWS_DIR", None)
        t0 = time.time()
        r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 1 and time.time() - t0 < 90, (r.returncode, r.stderr[-800:])
        assert f"rank {victim} was killed by signal 9" in r.stderr and "a rank failed" in r.stderr, r.stderr[-800:]
