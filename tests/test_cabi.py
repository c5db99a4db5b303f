"""CPU: the C-ABI library builds, loads, and exports every symbol that
include/fastlanes_amd.h declares; argument validation that needs no GPU works;
the product package never references oracle/."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build_library()
    import fastlanes_amd
    return fastlanes_amd.load()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "fastlanes_amd.h")).read()
    body = text.split("#define FL_DECLARE_TYPE(T, S)")[1].split("FL_DECLARE_TYPE(uint8_t, u8)")[0]
    per_type = re.findall(r"fl_##S##_(\w+)\(", body)
    syms = [f"fl_{ty}_{m}" for ty in ("u8", "u16", "u32", "u64") for m in per_type]
    # every other function either header names (a name followed by "(", in a declaration or in a comment about one)
    for h in ("fastlanes_amd.h", "fastlanes_amd_internal.h"):
        syms += re.findall(r"\b(fl_[a-z][a-z0-9_]*)\(", open(os.path.join(ROOT, "include", h)).read())
    return sorted(set(syms))


def test_every_declared_symbol_is_exported(lib):
    import fastlanes_amd
    syms = _header_symbols()
    assert len(syms) == 4 * 42 + 25
    assert sorted(fastlanes_amd.exported_symbols()) == syms
    for s in syms:
        assert hasattr(lib, s), s


def test_column_pair_and_bare_stream_argument_checks_need_no_gpu(lib):
    """fl_column_pair_alloc / fl_internal_bare_stream(_shape): every refusal happens before the first HIP call"""
    P = ctypes.c_void_p
    i, a, o, h = P(), P(), P(), P()
    kept = ctypes.c_int(-1)
    assert lib.fl_column_pair_alloc(1 << 20, 0, 1 << 20, 0, None, None, None, ctypes.byref(o), ctypes.byref(h), None, None) == 3   # FL_ERR_NULL
    assert lib.fl_column_pair_alloc(1 << 20, 128, 1 << 20, 0, None, ctypes.byref(i), None, ctypes.byref(o), ctypes.byref(h), None, None) == 3
    assert lib.fl_column_pair_alloc(1 << 20, 0, 1 << 20, 4, None, ctypes.byref(i), None, ctypes.byref(o), ctypes.byref(h), ctypes.byref(kept), None) == 2   # FL_LAYOUT_COUNT
    assert lib.fl_column_pair_alloc(1 << 20, 0, 1 << 20, -1, None, ctypes.byref(i), None, ctypes.byref(o), ctypes.byref(h), ctypes.byref(kept), None) == 2
    assert lib.fl_column_pair_free(None) == 0
    Z, I = ctypes.c_size_t, ctypes.c_int
    iu, au, ou, nt, wv, wn, bpu = Z(), Z(), Z(), I(), I(), I(), ctypes.c_uint()
    refs = [ctypes.byref(x) for x in (iu, au, ou, nt, wv, wn, bpu)]
    assert lib.fl_internal_bare_stream_shape(0, 32, 7, *refs) == 0
    assert (iu.value, au.value, ou.value, nt.value, wn.value) == (896, 0, 4096, 1, 31) and 3 <= wv.value <= 8     # u32 streams from W = 4 (round 6)
    assert lib.fl_internal_bare_stream_shape(0, 32, 3, *refs) == 0 and nt.value == 0
    assert lib.fl_internal_bare_stream_shape(0, 16, 7, *refs) == 0 and nt.value == 0 and lib.fl_internal_bare_stream_shape(0, 16, 8, *refs) == 0 and nt.value == 1
    assert lib.fl_internal_bare_stream_shape(1, 64, 17, *refs) == 0
    assert (iu.value, au.value, ou.value, nt.value, wn.value) == (8192, 0, 2176, 1, 16)
    assert lib.fl_internal_bare_stream_shape(2, 32, 12, *refs) == 0 and (iu.value, au.value, ou.value) == (1536, 128, 4096)
    assert lib.fl_internal_bare_stream_shape(3, 32, 33, *refs) == 0 and (iu.value, ou.value, nt.value, bpu.value) == (2112, 4096, 1, 1)
    assert lib.fl_internal_bare_stream_shape(0, 8, 3, *refs) == 0 and (iu.value, ou.value, bpu.value) == (4 * 384, 4096, 4)       # u8: 4 blocks per wavefront
    assert lib.fl_internal_bare_stream_shape(2, 16, 9, *refs) == 0 and (iu.value, au.value, ou.value, bpu.value) == (2 * 1152, 256, 4096, 2)
    assert lib.fl_internal_bare_stream_shape(0, 32, 33, *refs) == 1 and lib.fl_internal_bare_stream_shape(0, 12, 3, *refs) == 2
    assert lib.fl_internal_bare_stream_shape(4, 32, 3, *refs) == 2 and lib.fl_internal_bare_stream_shape(0, 32, 3, None, *refs[1:]) == 3
    p = ctypes.c_void_p(0x1000)
    assert lib.fl_internal_bare_stream(p, 896, None, 0, p, 4096, 0, 0, 5, 31, None) == 0        # nothing to do
    assert lib.fl_internal_bare_stream(None, 896, None, 0, p, 4096, 10, 0, 5, 31, None) == 3
    assert lib.fl_internal_bare_stream(ctypes.c_void_p(0x1008), 896, None, 0, p, 4096, 10, 0, 5, 31, None) == 4
    assert lib.fl_internal_bare_stream(p, 900, None, 0, p, 4096, 10, 0, 5, 31, None) == 2       # not a multiple of 16
    assert lib.fl_internal_bare_stream(p, 896, None, 0, p, 8208, 10, 0, 5, 31, None) == 2


def test_packed_len_and_strings(lib):
    assert lib.fl_packed_len(16, 3) == 192       # benches/bitpacking.rs:22
    assert lib.fl_packed_len(32, 10) == 320      # bitpacking.rs:251
    assert lib.fl_packed_len(64, 17) == 272
    assert lib.fl_packed_len(32, 33) == 0
    assert lib.fl_packed_len(12, 3) == 0
    assert lib.fl_status_string(1) == b"width > T"
    assert b"gfx950" in lib.fl_version()


def test_width_validation_needs_no_gpu(lib):
    buf = np.zeros(2048, dtype=np.uint64)
    p = buf.ctypes.data
    for ty, T in (("u8", 8), ("u16", 16), ("u32", 32), ("u64", 64)):
        assert getattr(lib, f"fl_{ty}_pack")(T + 1, p, p, 1, None) == 1
        assert getattr(lib, f"fl_{ty}_unpack")(T + 1, p, p, 1, None) == 1
        assert getattr(lib, f"fl_{ty}_undelta_pack")(T + 1, p, p, p, 1, None) == 1
        assert getattr(lib, f"fl_{ty}_unfor_pack")(T + 1, p, p, 1, p, 1, None) == 1
        assert getattr(lib, f"fl_{ty}_pack_host")(T + 1, p, p, 1) == 1
        assert getattr(lib, f"fl_{ty}_unpack")(3, None, p, 1, None) == 3         # FL_ERR_NULL
        assert getattr(lib, f"fl_{ty}_unpack")(3, p + 4, p, 1, None) == 4        # FL_ERR_ALIGN
        assert getattr(lib, f"fl_{ty}_unpack")(3, p, p, 0, None) == 0            # empty column
        v = (ctypes.c_uint64 * 1)()
        assert getattr(lib, f"fl_{ty}_unpack_single_host")(3, p, 1, 1024, v) == 2  # bitpacking.rs:152


def test_mixed_width_for_delta_validation_needs_no_gpu(lib):
    """fl_<ty>_*_widths (FoR / Delta over mixed-width columns) and fl_<ty>_for_widths: the argument checks that precede any launch."""
    buf = np.zeros(4096, dtype=np.uint64)
    p = buf.ctypes.data
    for ty in ("u8", "u16", "u32", "u64"):
        f = lambda m: getattr(lib, f"fl_{ty}_{m}")
        # empty column: nothing to do, whatever the pointers
        assert f("unfor_pack_widths")(None, None, None, 0, None, 1, None, 0, None, None) == 0
        assert f("for_pack_widths")(None, None, None, None, 1, None, 0, 0, None, None) == 0
        assert f("undelta_pack_widths")(None, None, None, 0, None, None, 0, None, None) == 0
        assert f("undelta_pack_untranspose_widths")(None, None, None, 0, None, None, 0, None, None) == 0
        assert f("transpose_delta_pack_widths")(None, None, None, None, None, 0, 0, None, None) == 0
        assert f("for_widths")(None, None, 0, None, None) == 0
        # FL_ERR_NULL: widths / offsets / references / bases / the data
        assert f("unfor_pack_widths")(None, p, p, 128, p, 1, p, 1, None, None) == 3
        assert f("unfor_pack_widths")(p, None, p, 128, p, 1, p, 1, None, None) == 3
        assert f("unfor_pack_widths")(p, p, p, 128, None, 1, p, 1, None, None) == 3
        assert f("unfor_pack_widths")(p, p, p, 128, p, 1, None, 1, None, None) == 3
        assert f("for_pack_widths")(p, p, None, p, 1, p, 128, 1, None, None) == 3
        assert f("for_pack_widths")(p, p, p, None, 1, p, 128, 1, None, None) == 3
        assert f("undelta_pack_widths")(p, p, p, 128, None, p, 1, None, None) == 3
        assert f("undelta_pack_widths")(p, None, p, 128, p, p, 1, None, None) == 3
        assert f("undelta_pack_widths")(p, p, None, 128, p, p, 1, None, None) == 3
        assert f("transpose_delta_pack_widths")(p, p, None, p, p, 128, 1, None, None) == 3
        assert f("transpose_delta_pack_widths")(p, p, p, p, None, 128, 1, None, None) == 3
        assert f("for_widths")(None, p, 1, p, None) == 3
        assert f("for_widths")(p, p, 1, None, None) == 3
        # FL_ERR_ALIGN: 16-byte columns and bases
        assert f("unfor_pack_widths")(p, p, p + 8, 128, p, 1, p, 1, None, None) == 4
        assert f("undelta_pack_widths")(p, p, p, 128, p + 8, p, 1, None, None) == 4
        assert f("transpose_delta_pack_widths")(p, p, p + 8, p, p, 128, 1, None, None) == 4


def test_internal_kernel_policy_is_validated(lib):
    """include/fastlanes_amd_internal.h: mode 0..2, waves 0 or 3..8 and blocks-per-wave 0..16 (mode 2 only), tile-map window 0 or
    8..31 (any mode); anything else resets to 0 -- a stray value must not select a kernel shape that was never tested."""
    try:
        for ok in (0, 1, 2, 2 + 256 * 3, 2 + 256 * 8, 2 + 65536 * 16, 2 + 256 * 6 + 65536 * 8, 2 + 65536 * 8 + (1 << 24),
                   16 << 25, 31 << 25, 8 << 25, 2 + 256 * 8 + (12 << 25)):      # bits 25-29: the tile-map window, any mode
            lib.fl_internal_set_kernel_policy(ok)
            assert lib.fl_internal_get_kernel_policy() == ok
        for bad in (-1, 3, 255, 2 + 256 * 2, 2 + 256 * 9, 2 + 65536 * 17, 1 + 256 * 4, 65536 * 2, 1 << 24, 2 + (1 << 24), 2 + 65536 * 4 + (2 << 24), 2 + (1 << 30),
                    1 << 25, 7 << 25):                                          # a window below 2^8 blocks
            lib.fl_internal_set_kernel_policy(bad)
            assert lib.fl_internal_get_kernel_policy() == 0, bad
    finally:
        lib.fl_internal_set_kernel_policy(0)


def test_fill_random_validation_needs_no_gpu(lib):
    buf = np.zeros(16, dtype=np.uint64)
    p = buf.ctypes.data
    assert lib.fl_fill_random(None, 0, 1, None) == 0            # empty
    assert lib.fl_fill_random(None, 64, 1, None) == 3           # FL_ERR_NULL
    assert lib.fl_fill_random(p + 4, 64, 1, None) == 4          # FL_ERR_ALIGN: 8-byte words
    assert lib.fl_fill_random(p, 60, 1, None) == 4
    assert lib.fl_status_string(6) == b"block outside the packed column"
    assert b"FL_CHECK_DEVICE" in lib.fl_status_string(7) and b"current device" in lib.fl_status_string(7)      # FL_ERR_DEVICE


def test_probe_memory_classes_validation_needs_no_gpu(lib):
    import ctypes
    out = (ctypes.c_int * 4)(7, 7, 7, 7)
    assert lib.fl_internal_probe_memory_classes(None, 0, out, None) == 0 and list(out) == [7, 7, 7, 7]          # no whole granule: nothing to say
    assert lib.fl_internal_probe_memory_classes(None, (8 << 30) - 1, out, None) == 0
    assert lib.fl_internal_probe_memory_classes(None, 16 << 30, out, None) == 3                                     # FL_ERR_NULL
    assert lib.fl_internal_probe_memory_classes(ctypes.c_void_p(0x1000), 16 << 30, None, None) == 3
    assert lib.fl_internal_probe_memory_classes(ctypes.c_void_p(0x1008), 16 << 30, out, None) == 4                # FL_ERR_ALIGN


def test_python_mirror_raises_like_the_reference():
    import fastlanes_amd as fl
    with pytest.raises(fl.FastLanesError):
        fl.BitPacking.pack(17, np.zeros(1024, dtype=np.uint16))
    with pytest.raises(ValueError):
        fl.BitPacking.pack(3, np.zeros(1000, dtype=np.uint16))


def test_python_mirror_checks_every_output_length():
    """A short caller-supplied `output` would be overrun by the kernel (it sizes its stores from n_blocks):
    every method checks it like the reference's length asserts (bitpacking.rs:78-80,111-113), before any call
    into the library (so this needs no GPU)."""
    import fastlanes_amd as fl
    v = np.zeros(2048, dtype=np.uint16)
    pk = np.zeros(2 * 192, dtype=np.uint16)
    base = np.zeros(2 * 64, dtype=np.uint16)
    short_un, short_pk = np.zeros(2047, dtype=np.uint16), np.zeros(191, dtype=np.uint16)
    calls = [
        lambda: fl.BitPacking.pack(3, v, output=short_pk),
        lambda: fl.BitPacking.unpack(3, pk, output=short_un),
        lambda: fl.FoR.for_pack(3, v, 5, output=short_pk),
        lambda: fl.FoR.unfor_pack(3, pk, 5, output=short_un),
        lambda: fl.Delta.delta(v, base, output=short_un),
        lambda: fl.Delta.undelta(v, base, output=short_un),
        lambda: fl.Delta.undelta_pack(3, pk, base, output=short_un),
        lambda: fl.Transpose.transpose(v, output=short_un),
        lambda: fl.Transpose.untranspose(v, output=short_un),
        lambda: fl.BitPacking.unpack(3, pk, output=np.zeros(2048, dtype=np.uint32)),      # wrong element type
    ]
    for i, c in enumerate(calls):
        with pytest.raises((ValueError, TypeError)):
            c()
    with pytest.raises(fl.FastLanesError):           # 300 must not wrap to 44 in a uint8 cast (bitpacking.rs:93)
        fl.MixedWidthPlan("u32", np.array([3, 300]))
    with pytest.raises(fl.FastLanesError):
        fl.MixedWidthPlan("u16", np.array([17], dtype=np.uint8))


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "fastlanes_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                for line in text.splitlines():
                    if "oracle" in line.lower():
                        stripped = line.strip()
                        assert stripped.startswith(("//", "#", "*", '"', "'")) or "oracle/" in stripped and (
                            "touches" in stripped or "never" in stripped.lower()), (f, line)
    header = open(os.path.join(ROOT, "include", "fastlanes_amd.h")).read()
    assert "fl_oracle" not in header


def test_cpp_trait_mirror_builds_and_fails_loudly_without_gpu(lib):
    """include/fastlanes_amd.hpp + tests/cpp/test_trait_mirror.cpp compile and link against the
    C ABI; with no GPU the reference's tests cannot silently pass on some CPU path."""
    import subprocess
    import torch
    exe = build_cpp_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if not torch.cuda.is_available():
        assert r.returncode == 2 and "HIP runtime error" in r.stdout, (r.returncode, r.stdout)


def build_cpp_test():
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "test_trait_mirror")
    src = exe + ".cpp"
    hdr = os.path.join(ROOT, "include", "fastlanes_amd.hpp")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                               "-I", "/opt/rocm/include", src,
                               "-L", os.path.join(ROOT, "fastlanes_amd"), "-lfastlanes_amd",
                               "-L", "/opt/rocm/lib", "-lamdhip64",
                               "-Wl,-rpath,$ORIGIN/../../fastlanes_amd", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_functor_api_example_builds():
    """examples/fused_dict_decode.hip (a user-written fused kernel on fl_device.hpp, the
    counterpart of the reference's exported unpack! macro) cross-compiles for gfx950."""
    import __graft_entry__ as ge
    sos = ge.build_examples()
    assert hasattr(ctypes.CDLL(sos["fused_dict_decode"]), "example_dict_unpack_u32_w8")
    assert hasattr(ctypes.CDLL(sos["iterate_running_max"]), "example_running_max_u32")
    assert hasattr(ctypes.CDLL(sos["fused_for_pack"]), "example_min_for_pack_u32")            # the pack_rows (pack!) counterpart
    assert os.access(sos["column_decode"], os.X_OK)          # the plain-C caller links against the C ABI


def test_python_mirror_rejects_mismatched_buffers():
    """Outputs are never silently converted or copied: wrong element size / non-contiguous
    arrays raise instead of writing the result into a temporary."""
    import fastlanes_amd as fl
    v = np.zeros(1024, dtype=np.uint16)
    with pytest.raises(TypeError):
        fl.BitPacking.pack(3, v, output=np.zeros(96, dtype=np.uint32))
    with pytest.raises(TypeError):
        fl.BitPacking.pack(3, np.zeros(1024, dtype=np.float32))
    with pytest.raises(ValueError):
        fl.BitPacking.pack(3, np.zeros(2048, dtype=np.uint16)[::2])
    with pytest.raises(TypeError):
        fl.Delta.delta(v, np.zeros(64, dtype=np.uint8))


def test_headers_are_plain_c_and_cxx17():
    """include/fastlanes_amd.h must be consumable by a C compiler (cgo / bindgen / ctypes users);
    the C++ mirror by a plain C++17 compiler without HIP."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-x", "c",
                           os.path.join(inc, "fastlanes_amd.h")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c++", "-I", inc,
                           os.path.join(inc, "fastlanes_amd.hpp")])


def test_zone_aware_placement_helper_arithmetic():
    """fastlanes_amd/placement.py: the input (+ aux) sits at offset 0 (reads like to stay inside one 64-GiB zone), the output is
    centred on the first 64-GiB multiple that leaves room for it (writes like to be split over two zones); nothing overlaps
    (no GPU needed: torch.empty on the meta device)."""
    import torch
    from fastlanes_amd import placement as pl
    Z = pl.ZONE_BYTES
    assert Z == 64 << 30
    GB = 10 ** 9
    for ib, ob, ab in ((9 * GB, 41 * GB, 0), (41 * GB, 9 * GB, 0), (15 * GB, 41 * GB, GB), (82 * GB, 22 * GB, 0), (22 * GB, 82 * GB, 0),
                       (1000, 4096, 128), (60 * GB, 61 * GB, 0)):
        slab, src, aux, dst = pl.column_pair(ib, ob, torch.device("meta"), aux_bytes=ab)
        assert (src.numel(), aux.numel(), dst.numel()) == (ib, ab, ob)
        s0, a0, d0 = src.storage_offset(), aux.storage_offset(), dst.storage_offset()
        assert s0 == 0 and d0 % 256 == 0 and (ab == 0 or (a0 % 256 == 0 and a0 >= ib))
        in_end = a0 + ab if ab else ib
        assert in_end <= d0 and d0 + ob <= slab.numel()                       # nothing overlaps, everything inside the slab
        k = (d0 + ob // 2 + Z // 2) // Z
        assert k >= 1 and d0 < k * Z < d0 + ob and abs(d0 + ob // 2 - k * Z) <= 512     # centred on a 64-GiB multiple ...
        assert k == 1 or (k - 1) * Z - ob // 2 < in_end                       # ... the first one that leaves room for the input
    with pytest.raises(ValueError):
        pl.column_pair(9 * Z, 4096, torch.device("meta"))


def test_constructed_layout_chooses_chunks_by_class(lib):
    """fl_internal_choose_chunks = the arrangement FL_LAYOUT_INTERLEAVED makes from a measured class map (a pure host function): the input
    inside ONE class; the output rotating through the other two classes (a write-dominated pair) or through all three, by POSITION -- the
    k-th chunk of the x-th eighth takes letter (x + k), so that the eight XCDs' write positions under the whole-column tile map cycle
    through the classes at every moment; a scarce class's positions go to the largest surplus, classes outside the rotation first; no
    chunk used twice; creation order where no class can hold the input (unpack u32 W=7 in A | out B/C 0.865, out
    A/B/C 0.860, out A/B 0.855, out B alone 0.80, out A 0.78 of the peak; pack u32 W=7 out A/B/C 0.859, out B/C 0.84)."""
    import random

    def choose(classes, n_in, n_out, out_classes=2):
        m = {"A": 0, "B": 1, "C": 2, "?": -1}
        c = (ctypes.c_int * len(classes))(*[m[x] for x in classes])
        o = (ctypes.c_int * (n_in + n_out))()
        k = lib.fl_internal_choose_chunks(c, len(classes), n_in, n_out, out_classes, o)
        idx = list(o[:k])
        assert len(set(idx)) == len(idx) and all(0 <= i < len(classes) for i in idx)
        return "".join(classes[i] for i in idx[:n_in]), "".join(classes[i] for i in idx[n_in:]), idx

    def positions(cout, t):
        """classes under the eight XCDs' write positions when every XCD is a fraction t through its eighth of the output"""
        n = len(cout)
        return [cout[min(n - 1, int((x + t) * n / 8))] for x in range(8)]

    # a balanced pool, write-dominated pair: the input in one class, the output over the other two, 4 + 4 under the eight positions at all times
    for n_out in (31, 39, 48, 27):
        cin, cout, _ = choose("AAAABBBBCCCC" * 10, 9, n_out)
        others = [x for x in "ABC" if x != cin[0]]
        assert len(set(cin)) == 1 and cout.count(cin[0]) <= n_out // 8 and abs(cout.count(others[0]) - cout.count(others[1])) <= 4, (cin, cout)
        for t in (0.0, 0.26, 0.5, 0.77, 0.99):
            ps = positions(cout, t)
            assert max(ps.count(x) for x in "ABC") <= 5, (n_out, t, cout, ps)
    # ... the same pool, read-dominated pair: all three classes under the eight positions at all times, none more than 4 times
    for n_in, n_out in ((31, 7), (39, 9), (20, 20), (16, 25)):
        cin, cout, _ = choose("AAAABBBBCCCC" * 14, n_in, n_out, 3)
        assert len(set(cin)) == 1 and len(set(cout)) == 3
        assert max(cout.count(x) for x in "ABC") - min(cout.count(x) for x in "ABC") <= 3
        for t in (0.0, 0.3, 0.6, 0.95):
            ps = positions(cout, t)
            assert len(set(ps)) == 3 and max(ps.count(x) for x in "ABC") <= 4, (n_in, n_out, t, cout, ps)
    # one class scarce (what a box handed out in round 6): left-overs of the input's class join in, no class carries more than two thirds
    cin, cout, _ = choose("B" * 38 + "C" * 50 + "AAA" + "C" * 14, 9, 39)
    assert len(set(cin)) == 1 and max(cout.count(x) for x in "ABC") <= 26 and len(set(cout)) == 3, (cin, cout)
    assert all(max(positions(cout, t).count(x) for x in "ABC") <= 6 for t in (0.0, 0.2, 0.4, 0.6, 0.8, 0.99)), cout     # the eight positions mix classes
    # two classes only
    cin, cout, _ = choose("A" * 80 + "B" * 16, 9, 39)
    assert set(cin) == {"A"} and cout.count("B") == 16 and all(2 <= positions(cout, t).count("B") <= 5 for t in (0.0, 0.25, 0.5, 0.75, 0.99)), cout
    # one class only / nothing classified / input larger than any class: creation order
    for classes in ("A" * 60, "?" * 60, "ABC" * 20):
        n_in = 30 if classes.startswith("ABC") else 9
        _, _, idx = choose(classes, n_in, 20)
        if classes != "A" * 60:
            assert idx == list(range(n_in + 20)), classes
    # a pool that is too small yields fewer indices, never a repeated one
    _, _, idx = choose("ABCABC", 3, 9)
    assert len(idx) <= 6
    # seeded fuzz: never a duplicate, the input single-class whenever some class can hold it, every index delivered while the pool has chunks
    rnd = random.Random(6)
    for _ in range(300):
        n = rnd.randint(8, 160)
        classes = "".join(rnd.choice("AAABBC?") for _ in range(n))
        n_in, n_out = rnd.randint(1, n // 3), rnd.randint(1, n // 2)
        cin, cout, idx = choose(classes, n_in, n_out, rnd.choice((2, 3)))
        assert len(idx) == n_in + n_out
        if any(classes.count(x) >= n_in for x in "ABC") and idx != list(range(n_in + n_out)):
            assert len(set(cin)) == 1
