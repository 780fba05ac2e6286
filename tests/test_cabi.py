"""The drop-in boundary: libbang.so loads, exports every symbol include/bang_c.h declares, and fails loudly
(never falls back to a CPU path) when no HIP device is present.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "bang_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|char\s*\*|const char\s*\*)\s+\*?(bang_[a-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ("bang_load_e", "bang_set_searchparams_e", "bang_alloc_e", "bang_init_e", "bang_query_e", "bang_free_e",
                 "bang_unload_e", "bang_load_c", "bang_query_c", "bang_k_front", "bang_k_back", "bang_k_rerank",
                 "bang_k_lut_build", "bang_k_filter", "bang_k_pqdist", "bang_k_parent", "bang_k_center_queries"):
        assert must in names
    assert len(names) >= 35


def test_library_exports_every_declared_symbol(libbang):
    for name in _declared_functions():
        assert hasattr(libbang, name), name


def test_cpp_class_symbols_are_exported(libbang):
    """bang.h's BANGSearch<T> for float / uint8_t / int8_t (reference bang.h:85-87) -- what test_driver links against."""
    import bang_amd
    out = subprocess.check_output(["nm", "-DC", "--defined-only", bang_amd.lib_path()], text=True)
    for t in ("float", "unsigned char", "signed char"):
        for meth in ("bang_load(char*)", "bang_alloc(int)", "bang_init(int)", "bang_set_searchparams(int, int, _DistFunc)",
                     "bang_free()", "bang_unload()"):
            assert f"BANGSearch<{t}>::{meth}" in out, (t, meth)
        assert f"BANGSearch<{t}>::bang_query(" in out
    assert os.path.exists(os.path.join(os.path.dirname(os.path.dirname(bang_amd.lib_path())), "bin", "bang_search"))


def test_gpu_kernels_are_compiled_for_gfx950(libbang):
    import bang_amd
    blob = open(bang_amd.lib_path(), "rb").read()
    assert b"gfx950" in blob and b"front_kernel" in blob and b"back_kernel" in blob and b"rerank_kernel" in blob


def test_baseline_search_instances_run_without_scratch(libbang, tmp_path):
    """The self-paced search-kernel instances of the BASELINE layouts (SIFT1B: 70 chunks, DEEP100M: 74, SIFT1M: 32; with and without the early
    code-row request) sit within a few registers of their budget -- 168 VGPRs for 12 waves, 128 for 16 -- and one more live value becomes
    scratch traffic inside the row reduce.  Read from the code object that was just built (kernel descriptors in the ELF notes)."""
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    objs = [os.path.join(ROOT, "bang-billion-scale-ann_amd", "lib", n) for n in ("bang_search.o", "bang_search_b.o")]
    if not all(os.path.exists(t) for t in tools) or not all(os.path.exists(o) for o in objs):
        pytest.skip("llvm binutils / the kernel objects are not here")

    def usage_of(obj, tag):
        fat, co = str(tmp_path / f"fat{tag}.bin"), str(tmp_path / f"dev{tag}.co")
        subprocess.run([tools[0], "--dump-section", f".hip_fatbin={fat}", obj, str(tmp_path / "unused.o")], check=True)
        subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
        out = {}
        for blk in notes.split(".name:")[1:]:
            name = blk.split()[0]
            m = re.match(r"_Z13search_kernelILi(\d+)ELi(\d+)ELb([01])ELi(\d+)ELb([01])ELb([01])EEv10SearchArgs$", name)
            if m:
                key = tuple(int(x) for x in m.groups())                      # (PSZ, NDW, ALIGNED, NHI, HOST, SPEC)
                out[key] = (int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk).group(1)), int(re.search(r"\.vgpr_count:\s*(\d+)", blk).group(1)))
        return out
    # the instances live in two translation units of the one source file (Makefile: part 0 under the ILP scheduling strategy -- the BASELINE
    # layouts --, part 1 under the default scheduler): every instance exactly once, the BASELINE self-paced ones in part 0
    usage, part1 = usage_of(objs[0], 0), usage_of(objs[1], 1)
    assert not (set(usage) & set(part1)) and len(usage) + len(part1) == 64, (len(usage), len(part1))
    assert all(k[1] >= 24 or (k[1] == 19 and not k[2]) or (k[4] and k[1] >= 16) for k in part1), sorted(part1)
    want = [(2, 18, 1, 58, 0, 0), (2, 18, 1, 58, 0, 1), (2, 19, 1, 22, 0, 0), (2, 19, 1, 22, 0, 1), (4, 8, 1, 0, 0, 0)]
    for key in want:
        assert key in usage, (key, sorted(usage)[:4])
        scratch, vgprs = usage[key]
        assert scratch == 0 and vgprs <= (168 if key[1] >= 16 else 128), (key, scratch, vgprs)
    assert not any(k[4] == 1 and k[5] == 1 for k in usage)                 # (no SPEC instance of the host-paced form)


def test_pq_layout_and_pivot_packing_host_side(libbang):
    from bang_amd import binding
    from bang_amd.synth import chunk_offsets
    assert binding.pq_layout(chunk_offsets(128, 32), 128, 32) == (4, 32)       # SIFT1M: 4 dims/chunk, 8 code dwords
    assert binding.pq_layout(chunk_offsets(128, 70), 128, 70) == (2, 72)       # SIFT1B: padded to 18 dwords
    assert binding.pq_layout(chunk_offsets(96, 74), 96, 74) == (2, 76)         # DEEP100M
    assert binding.pq_layout(chunk_offsets(256, 16), 256, 16)[0] == 0          # 16 dims/chunk -> LUT path
    off = chunk_offsets(10, 4)                                                  # sizes 3,3,2,2 -> psz 4
    piv = np.arange(256 * 10, dtype=np.float32).reshape(256, 10)
    psz, mp = binding.pq_layout(off, 10, 4)
    assert (psz, mp) == (4, 16)
    packed = binding.pack_pivots(piv, off, 10, 4, psz, mp)
    assert packed.shape == (16, 256, 4)
    assert list(packed[0, 3]) == [30, 31, 32, 0] and list(packed[2, 255]) == [2556, 2557, 0, 0]
    assert not packed[4:].any()


def test_exact_size_pivot_table_host_side(libbang):
    """bang_pack_pivots_ragged: [nhi][256][2] then [mp - nhi][256][1] floats, zero entries for the padding chunks; refused
    (nhi = 0) for layouts that are not 2,..,2,1,..,1 dims wide."""
    from bang_amd import binding
    from bang_amd.synth import chunk_offsets
    for D, m, nhi_want in ((128, 70, 58), (96, 74, 22)):
        off = chunk_offsets(D, m)
        piv = np.arange(256 * D, dtype=np.float32).reshape(256, D)
        psz, mp = binding.pq_layout(off, D, m)
        nhi, tab = binding.pack_pivots_ragged(piv, off, D, m, mp)
        assert nhi == nhi_want and tab.size % 4 == 0 and tab.size >= nhi * 512 + (mp - nhi) * 256 + 1
        assert tab.size * 4 < mp * 256 * psz * 4                          # smaller than the padded table
        hi = tab[: nhi * 512].reshape(nhi, 256, 2)
        lo = tab[nhi * 512: nhi * 512 + (mp - nhi) * 256].reshape(mp - nhi, 256)
        for c in (0, 1, nhi - 1):
            assert np.array_equal(hi[c], piv[:, off[c]:off[c] + 2])
        for c in (nhi, m - 1):
            assert np.array_equal(lo[c - nhi], piv[:, off[c]])
        assert not lo[m - nhi:].any() and not tab[nhi * 512 + (mp - nhi) * 256:].any()
    assert binding.pack_pivots_ragged(np.zeros((256, 128), np.float32), chunk_offsets(128, 32), 128, 32, 32) == (0, None)   # 4 dims/chunk
    assert binding.pack_pivots_ragged(np.zeros((256, 128), np.float32), chunk_offsets(128, 64), 128, 64, 64) == (0, None)   # all 2-dim: nothing to save


def test_search_kernel_lds_budget_arithmetic(libbang):
    """bang_search_supported(psz, mp, nhi, L) = waves per workgroup whose worklists fit the 160 KB of LDS beside the pivot table
    (host-side arithmetic, mirrors the launcher)."""
    f = libbang.bang_search_supported
    f.restype = C.c_int
    f.argtypes = [C.c_uint32] * 4
    assert f(4, 32, 0, 70) == 16 and f(4, 32, 0, 152) == 16 and f(4, 32, 0, 200) == 13 and 4 <= f(4, 32, 0, 512) <= 6   # SIFT1M layout: 128 KB table
    assert f(2, 72, 0, 152) < f(2, 72, 58, 152)                    # 128 dims in 70 chunks: padded 144 KB vs exact-size 128 KB
    # long code rows (>= 64 chunks): the self-paced instances are compiled for 12 waves of 168 VGPRs, and every wave keeps a 1 KB
    # staging area for the cooperative row fetch beside its worklist
    assert f(2, 76, 0, 152) <= 4 and f(2, 76, 22, 152) == 12       # 96 dims in 74 chunks: only the exact-size table (96 KB) leaves room
    assert f(2, 72, 58, 152) == 12 and f(2, 72, 58, 200) == 10     # SIFT1B layout: 130 KB table + 12 x (worklist + 1 KB)
    assert f(0, 5, 0, 100) == 0 and f(4, 32, 0, 513) == 0          # LUT path / L beyond MAX_L: no search kernel
    g = libbang.bang_ragged_supported
    g.restype = C.c_int
    g.argtypes = [C.c_uint32] * 4
    assert g(2, 72, 58, 70) == 1 and g(2, 76, 22, 74) == 1 and g(2, 32, 20, 30) == 0 and g(4, 32, 0, 32) == 0


def test_search_geometry_spreads_and_balances_a_batch(libbang):
    """bang_search_geometry (host-side arithmetic of the bang_k_search launch; 256 CUs on an MI355X, and where no device answers): a batch
    smaller than one wave-full per CU is spread over all CUs; one of one to two wave-fulls runs as two EQUAL rounds (self-paced form only);
    larger batches use every wave that fits."""
    f = libbang.bang_search_geometry
    f.restype = C.c_int
    f.argtypes = [C.c_uint32] * 7 + [C.c_int] + [C.POINTER(C.c_uint32)] * 4

    def geo(Q, host_paced=0, max_waves=0, layout=(2, 72, 58, 152)):
        wg, w, nctx, gs = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        assert f(*layout, Q, 0, max_waves, host_paced, C.byref(wg), C.byref(w), C.byref(nctx), C.byref(gs)) == 0
        return wg.value, w.value

    assert geo(100) == (100, 1) and geo(1250) == (256, 5) and geo(2500) == (256, 10) and geo(3072) == (256, 12)
    assert geo(3073) == (256, 7) and geo(3400) == (256, 7) and geo(4000) == (256, 8) and geo(5000) == (256, 10) and geo(6144) == (256, 12)
    assert geo(6145) == (256, 12) and geo(10000) == (256, 12)                          # three rounds and more: every wave that fits
    assert geo(5000, max_waves=8) == (256, 8) and geo(4000, max_waves=10) == (256, 8)  # (the cap first, then the balance under it)
    assert geo(5000, host_paced=1)[1] == geo(10000, host_paced=1)[1]                   # host-paced groups are not re-balanced
    assert geo(6000, layout=(4, 32, 0, 70)) == (256, 12) and geo(10000, layout=(4, 32, 0, 70)) == (256, 16)   # SIFT1M layout: 16 waves fit


def test_no_gpu_means_error_not_fallback(libbang):
    """On a box without a HIP device every engine entry point must fail with BANG_ERR_NOGPU (-6)."""
    import bang_amd
    if bang_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    e = bang_amd.Engine("uint8")
    with pytest.raises(bang_amd.BangError, match="no CPU fallback"):
        e.load("/nonexistent/prefix")
    p = C.c_void_p()
    assert libbang.bang_dev_malloc(C.byref(p), C.c_size_t(16)) == -6


def test_bad_arguments_are_rejected(libbang):
    import bang_amd
    e = bang_amd.Engine("float")
    with pytest.raises(bang_amd.BangError):
        e.set_searchparams(10, 5)            # L < k
    with pytest.raises(bang_amd.BangError):
        e.set_searchparams(10, 513)          # L > MAX_L (bang.h:20)
    with pytest.raises(bang_amd.BangError):
        e.alloc(100)                         # nothing loaded
    with pytest.raises(bang_amd.BangError):
        e.set_option("nonsense", 1)


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """The ctypes structures of bang_amd.binding are hand-written mirrors of include/bang_c.h: a C program compiled against the
    header prints sizeof and the offset of every field; size, field order and every offset must agree."""
    import re
    import shutil
    import subprocess
    from bang_amd import binding as B
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    pairs = {"bang_iter_params": B.IterParams, "bang_search_params": B.SearchParams, "bang_index_desc": B.IndexDesc,
             "bang_stats": B.Stats}
    hdr = open(os.path.join(ROOT, "include", "bang_c.h")).read()
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "bang_c.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct \{(.*?)\}\s*" + cname + r"\s*;", hdr, re.S)
        assert body, cname
        src.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for f, _ in cls._fields_:
            assert re.search(r"\b" + re.escape(f) + r"\b", body.group(1)), f"{cname}.{f} is not in the header"
            src.append(f'  printf("{cname} {f} %zu\\n", offsetof({cname}, {f}));')
    src += ['  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call([cc, "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(c)])
    got = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        s, f, v = line.split()
        got[(s, f)] = int(v)
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert got[(cname, f)] == getattr(cls, f).offset, f"{cname}.{f}"


def test_every_option_lives_in_one_table(libbang):
    """bang_describe_options prints the table csrc/bang_options.cpp defines; every key it lists is accepted by bang_set_option at
    both ends of its range and refused outside it, unknown keys are refused (no device needed: options are plain engine state)."""
    lib = libbang
    lib.bang_describe_options.argtypes = [C.c_char_p, C.c_size_t]
    need = lib.bang_describe_options(None, 0)
    buf = C.create_string_buffer(need)
    assert lib.bang_describe_options(buf, need) == need
    text = buf.value.decode()
    rows = re.findall(r"^  (\w+)\s+(BANG_\w+|-)\s+\[(-?\d+), (-?\d+)\]\s+(any time|bang_load|bang_alloc)", text, flags=re.M)
    keys = [r[0] for r in rows]
    for must in ("graph", "pull", "vectors", "lanes", "threads", "persistent", "search", "numa", "timing", "stage_zero_copy", "use_flag",
                 "compact", "fp_batch", "front_wgs", "check_every", "pq", "pq_ragged", "device", "host_walk_timeout_ms",
                 "kernel_go_timeout_ms"):
        assert must in keys, must
    for sw in ("BANG_PULL_ROWS_DIR", "BANG_STREAM_LOAD", "BANG_SEARCH_MAX_WAVES", "BANG_TIMELINE", "BANG_AMD_LIB"):
        assert sw in text
    h = C.c_void_p()
    assert lib.bang_create(0, C.byref(h)) == 0
    lib.bang_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    for key, env, lo, hi, _ in rows:
        lo, hi = int(lo), int(hi)
        assert lib.bang_set_option(h, key.encode(), lo) == 0, key
        assert lib.bang_set_option(h, key.encode(), hi) == 0, key
        if key not in ("pq_ragged", "use_flag", "compact"):      # flags take any value (non-zero = on)
            assert lib.bang_set_option(h, key.encode(), hi + 1) != 0, key
            assert lib.bang_set_option(h, key.encode(), lo - 1) != 0, key
    assert lib.bang_set_option(h, b"no_such_option", 1) != 0
    lib.bang_destroy.argtypes = [C.c_void_p]
    lib.bang_destroy(h)
    # every environment variable the engine sources read is in that table
    import glob
    known = set(re.findall(r"BANG_[A-Z_0-9]+", text))
    for src in glob.glob(os.path.join(ROOT, "bang-billion-scale-ann_amd", "csrc", "bang_*.cpp")):
        for name in re.findall(r'env_(?:long|flag|str)\("(BANG_[A-Z_0-9]+)"', open(src).read()):
            assert name in known, (src, name)
        if not src.endswith("bang_options.cpp"):
            assert "getenv(" not in open(src).read(), src


def test_integration_md_carries_the_option_table_verbatim(libbang):
    """INTEGRATION.md section 5 is the text bang_describe_options() prints: the documentation cannot drift from the table."""
    lib = libbang
    lib.bang_describe_options.argtypes = [C.c_char_p, C.c_size_t]
    need = lib.bang_describe_options(None, 0)
    buf = C.create_string_buffer(need)
    lib.bang_describe_options(buf, need)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert buf.value.decode().rstrip() in doc


def test_a_profiled_child_only_needs_an_up_to_date_library(monkeypatch):
    """binding.build() under BANG_NO_BUILD (what a bench child under rocprofv3 --pmc runs with: it must never start a compiler) refuses a
    stale LIBRARY -- but a harness source (csrc/test_driver*.cpp -> bin/*) edited after the last link of libbang.so does not make the library
    stale (round 5: it did, and every live PMC pass of bench.py failed with rc 1 on the GPU box)."""
    from bang_amd import binding
    real = os.path.getmtime
    newest = {"name": "test_driver.cpp"}

    def fake(p):
        return 4e9 if os.path.basename(p) == newest["name"] else real(p)
    monkeypatch.setattr(binding.os.path, "getmtime", fake)
    monkeypatch.setenv("BANG_NO_BUILD", "1")
    monkeypatch.delenv("BANG_AMD_LIB", raising=False)
    assert binding.build() == binding.lib_path()                    # a newer harness source: fine
    newest["name"] = "test_driver_multi.cpp"
    assert binding.build() == binding.lib_path()
    newest["name"] = "bang_lane.cpp"                                # a newer ENGINE source: the library is stale
    with pytest.raises(RuntimeError, match="stale"):
        binding.build()
    newest["name"] = "bang_c.h"
    with pytest.raises(RuntimeError, match="stale"):
        binding.build()


def test_no_build_product_is_tracked():
    """History stays source-only: no tracked file is an object, a library or a device-code bundle (round 5 had committed a stray
    `*.hipfb`, the offload bundle of a scratch file)."""
    import subprocess
    r = subprocess.run(["git", "ls-files"], cwd=ROOT, capture_output=True, text=True)
    if r.returncode != 0 or not r.stdout.strip():
        pytest.skip("not a git checkout (the GPU box gets a snapshot without .git)")
    bad = [f for f in r.stdout.splitlines() if f.endswith((".o", ".so", ".a", ".hipfb", ".hsaco", ".co", ".pyc"))]
    assert not bad, bad
