"""Committed golden fixture (tests/golden/, made by make_golden.py): the oracle must keep reproducing it and the
HIP path must match it bit for bit when the index is loaded from the reference-format FILES via bang_load."""
import os
import subprocess

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name, dtype):
    from bang_amd import formats
    ix = formats.read_index(os.path.join(GOLD, name), dtype)
    q = formats.read_bin(os.path.join(GOLD, name + "_query.bin"), dtype)
    return ix, q


def test_oracle_reproduces_golden_toy():
    from oracle import oracle as O
    exp = np.load(os.path.join(GOLD, "expected.npz"))
    ix, q = _load("toy", "float")
    assert (ix.N, ix.D, ix.R, ix.m, ix.medoid) == (14, 2, 4, 2, 6)
    ids, dists, st = O.Oracle(ix).search(q, 3, 4, with_stats=True)
    assert np.array_equal(ids, exp["toy_ids"]) and np.array_equal(dists.view(np.uint32), exp["toy_dists"].view(np.uint32))
    assert np.array_equal(st, exp["toy_stats"])


@pytest.mark.parametrize("L", [5, 17, 40])
def test_oracle_reproduces_golden_tiny(L):
    from oracle import oracle as O
    exp = np.load(os.path.join(GOLD, "expected.npz"))
    ix, q = _load("tiny", "uint8")
    ids, dists, st = O.Oracle(ix).search(q, 5, L, with_stats=True)
    assert np.array_equal(ids, exp[f"tiny_ids_L{L}"])
    assert np.array_equal(dists.view(np.uint32), exp[f"tiny_dists_L{L}"].view(np.uint32))
    assert np.array_equal(st, exp[f"tiny_stats_L{L}"])


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [0, 1])
def test_hip_path_reproduces_golden_from_files(libbang, graph):
    import bang_amd
    exp = np.load(os.path.join(GOLD, "expected.npz"))
    for name, dtype, k, Ls in (("toy", "float", 3, [4]), ("tiny", "uint8", 5, [5, 17, 40])):
        _, q = _load(name, dtype)
        with bang_amd.Engine(dtype, graph=graph) as e:
            e.load(os.path.join(GOLD, name))                      # bang_load: parses the four index files
            for L in Ls:
                e.set_searchparams(k, L)
                e.alloc(q.shape[0])
                e.init(q.shape[0])
                ids, dists = e.query(q)
                e.free()
                key = "toy" if name == "toy" else f"tiny"
                ids_e = exp["toy_ids"] if name == "toy" else exp[f"tiny_ids_L{L}"]
                d_e = exp["toy_dists"] if name == "toy" else exp[f"tiny_dists_L{L}"]
                assert np.array_equal(ids, ids_e), (name, L)
                assert np.array_equal(dists.view(np.uint32), d_e.view(np.uint32)), (name, L)
            e.unload()


@pytest.mark.gpu
def test_bang_search_cli_table(libbang):
    """The harness CLI (reference test_driver.cpp:564-599): auto sweep prints `L  Time  QPS  recall` rows, 5 per L."""
    import bang_amd
    exe = os.path.join(os.path.dirname(os.path.dirname(bang_amd.lib_path())), "bin", "bang_search")
    out = subprocess.run([exe, os.path.join(GOLD, "tiny"), os.path.join(GOLD, "tiny_query.bin"),
                          os.path.join(GOLD, "tiny_gt.bin"), "24", "5", "uint8", "l2", "auto"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l.split("\t") for l in out.stdout.splitlines() if l[:1].isdigit() and l.count("\t") == 3]
    assert "5-r@5" in out.stdout and "Bang Unload" in out.stdout
    Ls = sorted({int(r[0]) for r in rows})
    assert Ls[0] == 5 and Ls[1] == 17 and Ls[-1] <= 512 and all(b - a == 12 for a, b in zip(Ls, Ls[1:]))
    assert all(sum(1 for r in rows if int(r[0]) == L) == 5 for L in Ls)
    rec = {int(r[0]): float(r[3]) for r in rows}
    assert rec[5] < rec[41] and rec[Ls[-1]] >= 95.0


def test_bang_load_reports_missing_files(libbang):
    import bang_amd
    if bang_amd.device_count() == 0:
        pytest.skip("needs a HIP device to get past the no-GPU check")
    e = bang_amd.Engine("uint8")
    with pytest.raises(bang_amd.BangError, match="cannot open"):
        e.load("/nonexistent/prefix")
