"""The reference's own harness on top of our library (drop-in boundary).

oracle/_ref/ref_bang_search = /root/reference/BANG_Base/test_driver.cpp, UNMODIFIED, compiled against include/bang.h and
linked with lib/libbang.so (recipe: oracle/Makefile, target `ref`; built by __graft_entry__.build() wherever the reference
checkout is present -- the binary then travels to the GPU box with the other built artefacts).

* CPU box: the file compiles and links, i.e. include/bang.h + libbang.so provide every declaration and symbol the reference
  harness uses (bang.h:36-87).
* GPU box: the reference's main() / readers / calculate_recall (test_driver.cpp:43-93,238-272,338-557) drive our engine and
  print the same recall column as our own bin/bang_search on the same index."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/BANG_Base/test_driver.cpp"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_bang_search")
GOLD = os.path.join(ROOT, "tests", "golden")


def test_unmodified_reference_driver_compiles_and_links_against_libbang(libbang):
    if not os.path.exists(REF_SRC):
        pytest.skip("reference checkout not present (GPU box): the prebuilt oracle/_ref/ref_bang_search is used there")
    if os.path.exists(REF_BIN):
        os.remove(REF_BIN)
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(REF_BIN)
    und = subprocess.run(["nm", "-D", "--undefined-only", "-C", REF_BIN], capture_output=True, text=True).stdout
    need = [l for l in und.splitlines() if "BANGSearch<" in l]
    assert any("bang_query" in l for l in need) and any("bang_load" in l for l in need)
    defined = subprocess.run(["nm", "-D", "--defined-only", "-C", libbang._name], capture_output=True, text=True).stdout
    for l in need:                                     # every BANGSearch<T> member the reference harness calls is exported by us
        sym = l.split(" U ", 1)[1].strip()
        assert sym in defined, sym
    # without arguments it prints the reference's usage text and needs no GPU
    out = subprocess.run([REF_BIN], capture_output=True, text=True)
    assert "Too few parameters" in out.stdout + out.stderr


def _table(stdout):
    rows = [l.split("\t") for l in stdout.splitlines() if l[:1].isdigit() and l.count("\t") == 3]
    return [(int(r[0]), r[3].strip()) for r in rows]


@pytest.mark.gpu
@pytest.mark.parametrize("graph", ["host", "device"])
def test_reference_harness_drives_our_engine(libbang, graph):
    """Auto sweep (9 arguments, test_driver.cpp:384-418) of the reference binary and of ours on the `tiny` fixture: same L grid,
    5 rows per L, identical recall strings."""
    import bang_amd
    if not os.path.exists(REF_BIN):
        pytest.skip("oracle/_ref/ref_bang_search was not built (needs the reference checkout at build time)")
    ours = os.path.join(os.path.dirname(os.path.dirname(bang_amd.lib_path())), "bin", "bang_search")
    args = [os.path.join(GOLD, "tiny"), os.path.join(GOLD, "tiny_query.bin"), os.path.join(GOLD, "tiny_gt.bin"),
            "24", "5", "uint8", "l2", "auto"]
    env = dict(os.environ, BANG_GRAPH=graph)
    a = subprocess.run([REF_BIN] + args, capture_output=True, text=True, timeout=900, env=env)
    b = subprocess.run([ours] + args, capture_output=True, text=True, timeout=900, env=env)
    assert a.returncode == 0, a.stdout[-1500:] + a.stderr[-1500:]
    assert b.returncode == 0, b.stderr[-1500:]
    ta, tb = _table(a.stdout), _table(b.stdout)
    assert len(ta) >= 5 * 10 and ta == tb
    assert ta[0][0] == 5 and float(ta[-1][1]) >= 95.0
