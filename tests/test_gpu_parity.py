"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs, against the committed golden
fixtures, and -- at BASELINE.json's full sizes -- through size-independent
properties.  Bit-exact everywhere (integer work)."""
import json
import os

import numpy as np
import pytest

from datagen import sha, values
from golden.make_golden import N_BLOCKS, case_inputs
from oracle_lib import TYPES, lanes, packed_len, tbits

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))
TYS = ["u8", "u16", "u32", "u64"]


@pytest.fixture(scope="module")
def fl():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import fastlanes_amd
    fastlanes_amd.load()  # fails loudly if the HIP extension is missing
    return fastlanes_amd


def to_dev(a):
    import torch
    a = np.ascontiguousarray(a)
    if a.size == 0:
        return torch.empty(0, dtype=getattr(torch, str(a.dtype)), device="cuda:0")
    return torch.from_numpy(a.view(np.uint8)).to("cuda:0").view(getattr(torch, str(a.dtype)))


def to_np(t, ty):
    import torch
    if t.numel() == 0:
        return np.zeros(0, dtype=TYPES[ty][0])
    return t.view(torch.uint8).cpu().numpy().view(TYPES[ty][0])


# ---------------------------------------------------------------------------
# every (T, W): all width-parameterised ops vs the oracle, ragged block count
# ---------------------------------------------------------------------------
@pytest.fixture
def kernel_policy(fl):
    """fl_internal_set_kernel_policy for one test, restored afterwards (0 automatic, 1 cell-column kernels, 2 wave-per-block)."""
    lib = fl.load()

    def set_policy(p):
        lib.fl_internal_set_kernel_policy(p)
        assert lib.fl_internal_get_kernel_policy() == p
    yield set_policy
    lib.fl_internal_set_kernel_policy(0)


@pytest.mark.parametrize("policy", [0, 1, 2])
@pytest.mark.parametrize("ty", TYS)
def test_all_widths_vs_oracle(fl, oracle, kernel_policy, ty, policy):
    """Every (T, W) x {pack, unpack, for_pack, unfor_pack, undelta_pack} against the oracle -- under the automatic
    kernel choice and with each of the two kernel designs forced, so BOTH are parity-tested on all 124 pairs
    whatever fl_dispatch.hpp currently prefers."""
    kernel_policy(policy)
    T = tbits(ty)
    n = 37  # not a multiple of the 32-block workgroup, nor of the 8-block wavefront, nor of the 4-block one
    for w in range(T + 1):
        seed = 7000 + 64 * T + w
        v = values(ty, n * 1024, seed)                       # over-wide: pack must truncate
        pk = values(ty, n * packed_len(ty, w), seed + 1)     # arbitrary packed bits
        refs = values(ty, n, seed + 2)
        bases = values(ty, n * lanes(ty), seed + 3)
        dv, dpk, drefs, dbases = to_dev(v), to_dev(pk), to_dev(refs), to_dev(bases)

        got = to_np(fl.BitPacking.pack(w, dv), ty)
        assert np.array_equal(got, oracle.batch("pack", ty, w, v)), (ty, w, "pack")
        got = to_np(fl.BitPacking.unpack(w, dpk, n_blocks=n), ty)
        assert np.array_equal(got, oracle.batch("unpack", ty, w, pk, n_blocks=n)), (ty, w, "unpack")
        got = to_np(fl.FoR.for_pack(w, dv, drefs), ty)
        assert np.array_equal(got, oracle.batch("for_pack", ty, w, v, aux=refs)), (ty, w, "for_pack")
        got = to_np(fl.FoR.unfor_pack(w, dpk, drefs, n_blocks=n), ty)
        assert np.array_equal(got, oracle.batch("unfor_pack", ty, w, pk, aux=refs, n_blocks=n)), (ty, w, "unfor_pack")
        # one reference broadcast to every block (reference_stride 0): the scalar form of ffor.rs:5-17
        one = drefs[:1]
        got = to_np(fl.FoR.unfor_pack(w, dpk, one, n_blocks=n), ty)
        assert np.array_equal(got, oracle.batch("unfor_pack", ty, w, pk, aux=np.full(n, refs[0], dtype=refs.dtype), n_blocks=n)), (ty, w, "unfor_pack bcast")
        got = to_np(fl.Delta.undelta_pack(w, dpk, dbases), ty)
        assert np.array_equal(got, oracle.batch("undelta_pack", ty, w, pk, aux=bases, n_blocks=n)), (ty, w, "undelta_pack")


@pytest.mark.parametrize("waves", [0, 3, 8])
@pytest.mark.parametrize("ty", ["u32", "u64"])
def test_undelta_pack_two_blocks_per_wavefront(fl, oracle, kernel_policy, ty, waves):
    """Delta::undelta_pack::<W> (delta.rs:47-63) of the wide types through the TWO-blocks-per-wavefront form of the pipeline kernel
    (a dispatch-table entry 10 + k; forced here by the policy's blocks-per-wavefront field): every width against the oracle, an odd
    block count (the last wavefront holds ONE block), and byte for byte the same as the one-block form."""
    T = tbits(ty)
    n = 37
    for w in range(T + 1):
        seed = 7700 + 64 * T + w
        pk = values(ty, n * packed_len(ty, w), seed)
        bases = values(ty, n * lanes(ty), seed + 1)
        kernel_policy(2 + 256 * waves + 65536 * 2)
        two = to_np(fl.Delta.undelta_pack(w, to_dev(pk), to_dev(bases)), ty)
        assert np.array_equal(two, oracle.batch("undelta_pack", ty, w, pk, aux=bases, n_blocks=n)), (ty, w, waves)
        kernel_policy(2 + 256 * waves + 65536 * 1)
        assert np.array_equal(two, to_np(fl.Delta.undelta_pack(w, to_dev(pk), to_dev(bases)), ty)), (ty, w, "one block per wavefront")


@pytest.mark.parametrize("policy", [0, 1, 2])
@pytest.mark.parametrize("ty", TYS)
def test_delta_transpose_vs_oracle(fl, oracle, kernel_policy, ty, policy):
    kernel_policy(policy)
    n = 37
    v = values(ty, n * 1024, 91 + tbits(ty))
    bases = values(ty, n * lanes(ty), 92 + tbits(ty))
    dv, db = to_dev(v), to_dev(bases)
    assert np.array_equal(to_np(fl.Delta.delta(dv, db), ty), oracle.batch("delta", ty, None, v, aux=bases))
    assert np.array_equal(to_np(fl.Delta.undelta(dv, db), ty), oracle.batch("undelta", ty, None, v, aux=bases))
    assert np.array_equal(to_np(fl.Transpose.transpose(dv), ty), oracle.batch("transpose", ty, None, v))
    assert np.array_equal(to_np(fl.Transpose.untranspose(dv), ty), oracle.batch("untranspose", ty, None, v))


@pytest.mark.parametrize("ty", TYS)
def test_unpack_single_every_width_every_index(fl, oracle, ty):
    """The reference's try_round_trip (bitpacking.rs:273-315) checks unpack_single for EVERY index of
    every (T, W).  Same matrix here, two blocks per (T, W), one batched call each: against the oracle's
    closed-form reader (bitpacking.rs:132-179), against the oracle's unpack()[i], against the GPU's own
    unpack()[i] -- and the digest of the values against the committed golden fixture."""
    import torch
    T = tbits(ty)
    n = 2
    idx = np.arange(n * 1024, dtype=np.int64)
    didx = torch.from_numpy(idx).cuda()
    for w in range(T + 1):
        pl = packed_len(ty, w)
        pk = values(ty, n * pl, 3300 + 64 * T + w)
        dpk = to_dev(pk)
        got = to_np(fl.BitPacking.unpack_single(w, dpk, didx, n_blocks=n), ty)
        full = oracle.batch("unpack", ty, w, pk, n_blocks=n)
        assert np.array_equal(got, full), (ty, w, "unpack()[i]")
        assert np.array_equal(got, to_np(fl.BitPacking.unpack(w, dpk, n_blocks=n), ty)), (ty, w, "gpu unpack()[i]")
        for i in (0, 1, 15, 16, 127, 128, 1023, 1024, 1500, 2047):       # closed form, spot indices of both blocks
            b = i // 1024
            assert int(got[i]) == oracle.unpack_single(ty, w, pk[b * pl:(b + 1) * pl], i % 1024), (ty, w, i)
        assert sha(got) == GOLDEN["unpack_single"][f"{ty}_w{w}"], (ty, w, "golden")
    # closed form for every index at a few widths (the per-index ctypes loop is the slow part)
    for w in sorted({1, 3, T // 2 + 1, T - 1, T}):
        pl = packed_len(ty, w)
        pk = values(ty, pl, 3400 + w)
        got = to_np(fl.BitPacking.unpack_single(w, to_dev(pk), didx[:1024], n_blocks=1), ty)
        want = [oracle.unpack_single(ty, w, pk, i) for i in range(1024)]
        assert [int(x) for x in got] == want, (ty, w)
    # out-of-range index: the reference asserts (bitpacking.rs:152)
    with pytest.raises(fl.FastLanesError):
        fl.BitPacking.unpack_single(3, to_dev(values(ty, packed_len(ty, 3), 1)),
                                    torch.tensor([1024], dtype=torch.int64).cuda(), n_blocks=1)


# ---------------------------------------------------------------------------
# committed golden fixtures (no oracle involved at run time)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("policy", [1, 2])
@pytest.mark.parametrize("ty", TYS)
def test_golden_fixtures(fl, kernel_policy, ty, policy):
    kernel_policy(policy)
    T = tbits(ty)
    for w in range(T + 1):
        i = case_inputs(ty, w)
        g = GOLDEN["cases"][f"{ty}/{w}"]
        dv, dpk, drefs, dbases = (to_dev(i[k]) for k in ("values", "packed", "refs", "bases"))
        assert sha(to_np(fl.BitPacking.pack(w, dv), ty)) == g["pack"], (ty, w)
        assert sha(to_np(fl.BitPacking.unpack(w, dpk, n_blocks=N_BLOCKS), ty)) == g["unpack"], (ty, w)
        assert sha(to_np(fl.FoR.for_pack(w, dv, drefs), ty)) == g["for_pack"], (ty, w)
        assert sha(to_np(fl.FoR.unfor_pack(w, dpk, drefs, n_blocks=N_BLOCKS), ty)) == g["unfor_pack"], (ty, w)
        assert sha(to_np(fl.Delta.undelta_pack(w, dpk, dbases), ty)) == g["undelta_pack"], (ty, w)
    i = case_inputs(ty, T)
    g = GOLDEN["cases"][f"{ty}/misc"]
    dv, dbases = to_dev(i["values"]), to_dev(i["bases"])
    assert sha(to_np(fl.Delta.delta(dv, dbases), ty)) == g["delta"]
    assert sha(to_np(fl.Delta.undelta(dv, dbases), ty)) == g["undelta"]
    assert sha(to_np(fl.Transpose.transpose(dv), ty)) == g["transpose"]
    assert sha(to_np(fl.Transpose.untranspose(dv), ty)) == g["untranspose"]


def test_survey_known_answer_vectors(fl):
    k = GOLDEN["survey_kats"]
    v = (np.arange(1024) % 8).astype(np.uint16)
    assert sha(to_np(fl.BitPacking.pack(3, to_dev(v)), "u16")) == k["KAT-2 u16 W=3 v[i]=i%8"]
    v = np.arange(1024, dtype=np.uint32)
    assert sha(to_np(fl.BitPacking.pack(10, to_dev(v)), "u32")) == k["KAT-3 u32 W=10 v[i]=i"]
    v = (np.arange(1024) & 127).astype(np.uint32)
    assert sha(to_np(fl.BitPacking.pack(7, to_dev(v)), "u32")) == k["KAT-4 u32 W=7 v[i]=i&127"]
    v = np.array([(i * 2654435761) & 0x1FFFF for i in range(1024)], dtype=np.uint64)
    assert sha(to_np(fl.BitPacking.pack(17, to_dev(v)), "u64")) == k["KAT-5 u64 W=17 v[i]=(i*2654435761)&0x1FFFF"]
    # KAT-7: benches/delta.rs:15-27  transpose -> delta(base 0) -> pack W=9, fused decode returns transposed
    v = (np.arange(1024) // 8).astype(np.uint16)
    t = fl.Transpose.transpose(to_dev(v))
    base = to_dev(np.zeros(64, dtype=np.uint16))
    pk = fl.BitPacking.pack(9, fl.Delta.delta(t, base))
    assert sha(to_np(pk, "u16")) == k["KAT-7 u16 W=9 delta bench"]
    import torch
    assert torch.equal(fl.Delta.undelta_pack(9, pk, base).view(torch.int16), t.view(torch.int16))
    assert np.array_equal(to_np(fl.Transpose.untranspose(t), "u16"), v)


# ---------------------------------------------------------------------------
# reference unit tests, run against the GPU path through the host tier
# ---------------------------------------------------------------------------
def test_readme_example_host_tier(fl):
    # README.md:14-47 / lib.rs:71-96
    W = 3
    v = np.array([i % (1 << W) for i in range(1024)], dtype=np.uint16)
    pk = fl.BitPacking.unchecked_pack(W, v)
    assert pk.size == 128 * W // 2
    assert np.array_equal(fl.BitPacking.unchecked_unpack(W, pk), v)
    for i in range(0, 1024, 41):
        assert fl.BitPacking.unchecked_unpack_single(W, pk, i) == v[i]
    with pytest.raises(fl.FastLanesError):
        fl.BitPacking.unpack_single(W, pk, 1024)


def test_reference_ffor_and_delta_tests_host_tier(fl):
    # ffor.rs:66-88
    W = 15
    v = np.array([i % (1 << W) for i in range(1024)], dtype=np.uint16)
    pk = fl.FoR.for_pack(W, v, 10)
    assert np.array_equal(fl.BitPacking.unpack(W, pk), (v - np.uint16(10)) & np.uint16((1 << W) - 1))
    assert np.array_equal(fl.FoR.unfor_pack(W, pk, 10), ((v - np.uint16(10)) & np.uint16(0x7FFF)) + np.uint16(10))
    # delta.rs:80-107
    v = (np.arange(1024) // 8).astype(np.uint16)
    t = fl.Transpose.transpose(v)
    zero = np.zeros(64, dtype=np.uint16)
    pk = fl.BitPacking.pack(W, fl.Delta.delta(t, zero))
    assert np.array_equal(fl.Delta.undelta_pack(W, pk, zero), t)
    assert np.array_equal(fl.Delta.undelta(fl.BitPacking.unpack(W, pk), zero), t)


@pytest.mark.parametrize("ty", TYS)
def test_edge_cases(fl, ty):
    import torch
    T = tbits(ty)
    dt = getattr(torch, {"u8": "uint8", "u16": "uint16", "u32": "uint32", "u64": "uint64"}[ty])
    empty = torch.empty(0, dtype=dt, device="cuda:0")
    assert fl.BitPacking.pack(3, empty).numel() == 0                      # empty column
    assert fl.BitPacking.unpack(3, empty).numel() == 0
    with pytest.raises(fl.FastLanesError):                                 # bitpacking.rs:93
        fl.BitPacking.pack(T + 1, torch.zeros(1024, dtype=dt, device="cuda:0"))
    with pytest.raises(ValueError):                                        # bitpacking.rs:79
        fl.BitPacking.pack(3, torch.zeros(1000, dtype=dt, device="cuda:0"))
    # width 0: pack writes nothing, unpack zero-fills (macros.rs:52-53,118-125)
    out = torch.full((2048,), 1, dtype=torch.uint8, device="cuda:0").repeat(T // 8).view(dt)
    fl.BitPacking.unpack(0, empty, output=out, n_blocks=2)
    assert not out.view(torch.uint8).any()
    # single block, and exactly one workgroup / one wavefront worth of blocks
    for n in (1, 8, 32, 33):
        v = values(ty, n * 1024, n, bits=T - 1)
        d = to_dev(v)
        rt = fl.BitPacking.unpack(T - 1, fl.BitPacking.pack(T - 1, d))
        assert np.array_equal(to_np(rt, ty), v)


# ---------------------------------------------------------------------------
# BASELINE.json full sizes, through size-independent properties
# ---------------------------------------------------------------------------
def _rand_dev(nbytes, seed):
    import torch
    g = torch.Generator(device="cuda:0")
    g.manual_seed(seed)
    return torch.randint(-2**63, 2**63 - 1, (nbytes // 8,), dtype=torch.int64, device="cuda:0", generator=g)


SUBSET = 65_536


def _sample_blocks(n):
    return sorted({0, 1, 31, 32, n // 2, n - 33, n - 2, n - 1})


def test_config2_u32_w7_10M_blocks(fl, oracle):
    """u32 W=7 unpack, 10 M blocks: pack(unpack(x)) == x for EVERY packed x (all bit
    patterns are valid packed data), plus oracle comparison of sampled blocks."""
    import torch
    n = 10_000_000
    pk = _rand_dev(n * 896, 42).view(torch.uint32)
    out = fl.BitPacking.unpack(7, pk)
    assert out.numel() == n * 1024
    assert int(out.view(torch.int32).max()) <= 127 and int(out.view(torch.int32).min()) >= 0
    back = fl.BitPacking.pack(7, out)
    assert torch.equal(back.view(torch.int32), pk.view(torch.int32))
    for b in _sample_blocks(n):
        got = to_np(out[b * 1024:(b + 1) * 1024], "u32")
        assert np.array_equal(got, oracle.unpack("u32", 7, to_np(pk[b * 224:(b + 1) * 224], "u32"))), b
    # a >= 65 536-block subset through the CPU oracle, byte for byte
    for b0 in (0, n - SUBSET):
        want = oracle.fast("unpack", "u32", 7, to_np(pk[b0 * 224:(b0 + SUBSET) * 224], "u32"), nthreads=8)
        assert np.array_equal(to_np(out[b0 * 1024:(b0 + SUBSET) * 1024], "u32"), want)


def test_config3_u64_w17_10M_blocks(fl, oracle):
    import torch
    n = 10_000_000
    pk = _rand_dev(n * 2176, 43).view(torch.uint64)
    out = fl.BitPacking.unpack(17, pk)
    back = fl.BitPacking.pack(17, out)
    assert torch.equal(back.view(torch.int64), pk.view(torch.int64))
    assert int(out.view(torch.int64).max()) < (1 << 17) and int(out.view(torch.int64).min()) >= 0
    for b in _sample_blocks(n):
        got = to_np(out[b * 1024:(b + 1) * 1024], "u64")
        assert np.array_equal(got, oracle.unpack("u64", 17, to_np(pk[b * 272:(b + 1) * 272], "u64"))), b
    for b0 in (0, n - SUBSET):
        want = oracle.fast("unpack", "u64", 17, to_np(pk[b0 * 272:(b0 + SUBSET) * 272], "u64"), nthreads=8)
        assert np.array_equal(to_np(out[b0 * 1024:(b0 + SUBSET) * 1024], "u64"), want)
        wantp = oracle.fast("pack", "u64", 17, want, nthreads=8)
        assert np.array_equal(to_np(back[b0 * 272:(b0 + SUBSET) * 272], "u64"), wantp)


def test_config4_fused_delta_u32_w12_10M_blocks(fl, oracle):
    import torch
    n = 10_000_000
    pk = _rand_dev(n * 1536, 44).view(torch.uint32)
    bases = _rand_dev(n * 128, 45).view(torch.uint32)
    fused = fl.Delta.undelta_pack(12, pk, bases)
    unfused = fl.Delta.undelta(fl.BitPacking.unpack(12, pk), bases)           # benches/delta.rs:29-43
    assert torch.equal(fused.view(torch.int32), unfused.view(torch.int32))
    del unfused
    # delta() inverts it, and re-packing the deltas gives the input back
    back = fl.BitPacking.pack(12, fl.Delta.delta(fused, bases))
    assert torch.equal(back.view(torch.int32), pk.view(torch.int32))
    for b in _sample_blocks(n):
        got = to_np(fused[b * 1024:(b + 1) * 1024], "u32")
        want = oracle.undelta_pack("u32", 12, to_np(pk[b * 384:(b + 1) * 384], "u32"),
                                   to_np(bases[b * 32:(b + 1) * 32], "u32"))
        assert np.array_equal(got, want), b
    for b0 in (0, n - SUBSET):
        want = oracle.fast("undelta_pack", "u32", 12, to_np(pk[b0 * 384:(b0 + SUBSET) * 384], "u32"),
                           aux=to_np(bases[b0 * 32:(b0 + SUBSET) * 32], "u32"), nthreads=8)
        assert np.array_equal(to_np(fused[b0 * 1024:(b0 + SUBSET) * 1024], "u32"), want)


def test_cpp_trait_mirror_reference_tests(fl):
    """The reference's unit tests written against include/fastlanes_amd.hpp (C++ mirror of the
    traits), run on the GPU through the host tier."""
    import subprocess
    from test_cabi import build_cpp_test
    r = subprocess.run([build_cpp_test()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)


# ---------------------------------------------------------------------------
# mixed-width columns (BASELINE.json config 5)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("ty", TYS)
def test_mixed_width_plan_vs_oracle(fl, oracle, ty):
    import torch
    T = tbits(ty)
    rng = np.random.default_rng(77 + T)
    n = 300
    for widths in ((np.arange(n) % (T + 1)).astype(np.uint8),           # every width incl. 0 and T
                   rng.integers(0, T + 1, size=n).astype(np.uint8),     # random
                   np.full(n, 3, dtype=np.uint8)):                      # uniform (one bucket)
        esz = T // 8
        total = int(widths.astype(np.int64).sum()) * 128 // esz
        col = values(ty, total, 900 + T)
        plan = fl.MixedWidthPlan(ty, widths)
        assert plan.n_blocks == n and plan.packed_bytes == total * esz
        got = to_np(plan.unpack(to_dev(col)), ty)
        want, pos = [], 0
        for w in widths:
            k = packed_len(ty, int(w))
            want.append(oracle.unpack(ty, int(w), col[pos:pos + k]))
            pos += k
        want = np.concatenate(want)
        assert np.array_equal(got, want)
        # pack back: packing the decoded values reproduces the packed column bit for bit
        back = to_np(plan.pack(to_dev(want)), ty)
        assert np.array_equal(back, col)
        plan.close()
    with pytest.raises(fl.FastLanesError):                               # bitpacking.rs:93
        fl.MixedWidthPlan(ty, np.array([T + 1], dtype=np.uint8))
    # an index-less device name means the current device (torch.device('cuda') != torch.device('cuda:0'))
    for name in ("cuda", "cuda:0", torch.device("cuda")):
        plan = fl.MixedWidthPlan(ty, np.full(4, 3, dtype=np.uint8), device=name)
        pk = to_dev(values(ty, 4 * packed_len(ty, 3), 12))
        assert plan.unpack(pk).device == torch.device("cuda", 0)
        assert torch.equal(plan.pack(plan.unpack(pk)).view(torch.uint8), pk.view(torch.uint8))
        plan.close()
    with pytest.raises(ValueError):
        fl.MixedWidthPlan(ty, np.full(4, 3, dtype=np.uint8), device="cpu")


@pytest.mark.parametrize("ty", TYS)
def test_unpack_pack_widths_device_resident(fl, oracle, ty):
    """SURVEY 8(b) surface: widths[n_blocks] (u8) and offsets[n_blocks] (u64 byte offsets) live in HBM and are
    read by the kernel; nothing is built on the host.  Against the oracle's per-block loop
    (bitpacking.rs:76-96,109-129): every width 0..T, ragged block counts, over-wide pack inputs,
    offsets with gaps / in a permuted order, and the device-side error flag for a width > T."""
    import torch
    T = tbits(ty)
    esz = T // 8
    tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
    rng = np.random.default_rng(4242 + T)
    for n, widths in ((T + 1, np.arange(T + 1)), (1, np.array([T // 2])), (263, rng.integers(0, T + 1, size=263))):
        widths = widths.astype(np.uint8)
        dw = torch.from_numpy(widths).cuda()
        doff, dtotal = fl.widths_to_offsets(ty, dw)
        off = np.concatenate([[0], np.cumsum(widths.astype(np.int64) * 128)])
        assert np.array_equal(doff.cpu().numpy(), off[:-1]) and int(dtotal.item()) == off[-1]
        col = values(ty, int(off[-1]) // esz, 5100 + n)
        dcol = to_dev(col)
        got = to_np(fl.unpack_widths(dw, doff, dcol), ty)
        want = np.concatenate([oracle.unpack(ty, int(w), col[off[b] // esz:off[b + 1] // esz]) for b, w in enumerate(widths)])
        assert np.array_equal(got, want), (ty, n, "unpack_widths")
        # pack: over-wide inputs must be truncated exactly like pack::<W> does (macros.rs:73; not for W == T, :58)
        v = values(ty, n * 1024, 5200 + n)
        dpk = torch.full((int(off[-1]) // esz,), 0, dtype=tdt, device="cuda:0")
        fl.pack_widths(dw, doff, to_dev(v), dpk)
        wantp = np.concatenate([oracle.pack(ty, int(w), v[b * 1024:(b + 1) * 1024]) for b, w in enumerate(widths)]
                               + [np.zeros(0, dtype=TYPES[ty][0])])
        assert np.array_equal(to_np(dpk, ty), wantp), (ty, n, "pack_widths")
    # offsets are honoured as given: blocks in reverse order with a 256-byte gap after each
    n = 40
    widths = rng.integers(1, T + 1, size=n).astype(np.uint8)
    sizes = widths.astype(np.int64) * 128 + 256
    off = (np.cumsum(sizes[::-1])[::-1] - sizes).astype(np.int64)          # block 0 last
    total = int(sizes.sum())
    col = values(ty, total // esz, 5300)
    dw, doff = torch.from_numpy(widths).cuda(), torch.from_numpy(off).cuda()
    got = to_np(fl.unpack_widths(dw, doff, to_dev(col)), ty)
    want = np.concatenate([oracle.unpack(ty, int(w), col[off[b] // esz:off[b] // esz + packed_len(ty, int(w))])
                           for b, w in enumerate(widths)])
    assert np.array_equal(got, want)
    v = values(ty, n * 1024, 5301)
    guard = torch.full((total // esz,), 0x5A if ty == "u8" else 0x5A5A, dtype=tdt, device="cuda:0")
    before = to_np(guard, ty).copy()
    fl.pack_widths(dw, doff, to_dev(v), guard)
    after = to_np(guard, ty)
    for b, w in enumerate(widths):
        lo = off[b] // esz
        k = packed_len(ty, int(w))
        assert np.array_equal(after[lo:lo + k], oracle.pack(ty, int(w), v[b * 1024:(b + 1) * 1024])), b
        assert np.array_equal(after[lo + k:lo + k + 256 // esz], before[lo + k:lo + k + 256 // esz]), "gap bytes were written"
    # unpack_single over the same reversed / gapped column: every element of a few blocks, and random ones, vs unpack()
    idx = np.concatenate([np.arange(3 * 1024), rng.integers(0, n * 1024, size=5000), [n * 1024 - 1]]).astype(np.int64)
    got1 = to_np(fl.unpack_single_widths(dw, doff, to_dev(col), torch.from_numpy(idx).cuda()), ty)
    assert np.array_equal(got1, want[idx]), (ty, "unpack_single_widths")
    with pytest.raises(fl.FastLanesError):                                    # bitpacking.rs:152
        fl.unpack_single_widths(dw, doff, to_dev(col), torch.tensor([n * 1024], dtype=torch.int64).cuda())
    # width > T: that block is skipped and the flag raised (bitpacking.rs:93,126 unreachable!())
    bad = widths.copy()
    bad[7] = T + 1
    dbad = torch.from_numpy(bad).cuda()
    with pytest.raises(fl.FastLanesError):
        fl.unpack_widths(dbad, doff, to_dev(col))
    with pytest.raises(fl.FastLanesError):
        fl.widths_to_offsets(ty, dbad)
    with pytest.raises(fl.FastLanesError):                                    # bitpacking.rs:197
        fl.unpack_single_widths(dbad, doff, to_dev(col), torch.tensor([7 * 1024 + 5], dtype=torch.int64).cuda())
    out = torch.zeros(n * 1024, dtype=tdt, device="cuda:0")
    fl.unpack_widths(dbad, doff, to_dev(col), output=out, check=False)     # asynchronous form: no flag read-back
    g = to_np(out, ty)
    assert np.array_equal(g[:7 * 1024], want[:7 * 1024]) and np.array_equal(g[8 * 1024:], want[8 * 1024:])
    assert not g[7 * 1024:8 * 1024].any()
    # the kernels check every block's preconditions themselves (include/fastlanes_amd.h FL_DEVERR_*): a misaligned offset
    # (not a multiple of 16) or a block that does not lie inside the packed column is SKIPPED and flagged -- nothing is
    # read or written for it, every other block is processed (bitpacking.rs:78-80,111-113 debug_asserts, device-side)
    for what, status, mutate in (("misaligned", 4, lambda o: o.__setitem__(5, o[5] + 8)),
                                 ("past the end", 6, lambda o: o.__setitem__(5, total - 64)),
                                 ("far outside", 6, lambda o: o.__setitem__(5, 1 << 40))):
        boff = off.copy()
        mutate(boff)
        dboff = torch.from_numpy(boff).cuda()
        with pytest.raises(fl.FastLanesError) as ei:
            fl.unpack_widths(dw, dboff, to_dev(col))
        assert ei.value.status == status, what
        out = torch.zeros(n * 1024, dtype=tdt, device="cuda:0")
        fl.unpack_widths(dw, dboff, to_dev(col), output=out, check=False)
        g = to_np(out, ty)
        assert np.array_equal(g[:5 * 1024], want[:5 * 1024]) and np.array_equal(g[6 * 1024:], want[6 * 1024:]), what
        assert not g[5 * 1024:6 * 1024].any(), what
        guard2 = torch.full((total // esz,), 0x5A if ty == "u8" else 0x5A5A, dtype=tdt, device="cuda:0")
        with pytest.raises(fl.FastLanesError) as ei:
            fl.pack_widths(dw, dboff, to_dev(v), guard2)
        assert ei.value.status == status, what
        after2 = to_np(guard2, ty)
        lo5, k5 = off[5] // esz, packed_len(ty, int(widths[5]))
        keep = np.ones(after.size, dtype=bool)          # block 5 was written nowhere; every other byte as in the good run
        keep[lo5:lo5 + k5] = False
        assert np.array_equal(after2[keep], after[keep]), what
        assert np.array_equal(after2[lo5:lo5 + k5], before[lo5:lo5 + k5]), what
        if what != "misaligned":     # a lookup only needs element alignment
            with pytest.raises(fl.FastLanesError) as ei:
                fl.unpack_single_widths(dw, dboff, to_dev(col), torch.tensor([5 * 1024 + 9], dtype=torch.int64).cuda())
            assert ei.value.status == 6, what
    # an undersized packed tensor: the blocks that do not fit are skipped, not read out of bounds
    short = to_dev(col[:(total - int(sizes[0])) // esz + 1])                # block 0 lives at the END of the reversed column
    with pytest.raises(fl.FastLanesError) as ei:
        fl.unpack_widths(dw, doff, short)
    assert ei.value.status == 6
    # indices must be integer tensors of the same device (a float64 tensor has 8-byte elements too)
    with pytest.raises(TypeError):
        fl.unpack_single_widths(dw, doff, to_dev(col), torch.tensor([1.0], dtype=torch.float64).cuda())
    with pytest.raises(TypeError):
        fl.unpack_single_widths(dw, doff, to_dev(col), torch.tensor([1], dtype=torch.int64))
    # a column of width-0 blocks only has no packed bytes at all (a NULL packed pointer): zeros out, nothing packed
    z = torch.zeros(3, dtype=torch.uint8, device="cuda:0")
    zoff, ztot = fl.widths_to_offsets(ty, z)
    assert int(ztot.item()) == 0
    zo = fl.unpack_widths(z, zoff, torch.empty(0, dtype=tdt, device="cuda:0"))
    assert zo.numel() == 3 * 1024 and not to_np(zo, ty).any()
    fl.pack_widths(z, zoff, to_dev(values(ty, 3 * 1024, 1)), torch.empty(0, dtype=tdt, device="cuda:0"))
    assert not to_np(fl.unpack_single_widths(z, zoff, torch.empty(0, dtype=tdt, device="cuda:0"),
                                             torch.tensor([0, 1500, 3071], dtype=torch.int64).cuda()), ty).any()
    # empty column
    e8 = torch.empty(0, dtype=torch.uint8, device="cuda:0")
    o0, t0 = fl.widths_to_offsets(ty, e8)
    assert o0.numel() == 0 and int(t0.item()) == 0
    assert fl.unpack_widths(e8, o0, torch.empty(0, dtype=tdt, device="cuda:0")).numel() == 0


@pytest.mark.parametrize("waves", [0, 4, 8])
@pytest.mark.parametrize("ty", TYS)
def test_for_and_delta_over_mixed_width_columns(fl, oracle, kernel_policy, ty, waves):
    """FoR's and Delta's bodies over a device-resident mixed-width column: the reference's const-W methods called with block
    b's width (ffor.rs:24-50, delta.rs:47-63) -- per block against the oracle, every width 0..T, ragged counts, references
    per block and broadcast, the fused transpose extensions, the per-block device checks, and the encoder chain
    block_min_max -> for_widths -> widths_to_offsets -> for_pack_widths -> unfor_pack_widths as a lossless round trip."""
    import torch
    kernel_policy(2 + 256 * waves if waves else 0)
    T = tbits(ty)
    esz = T // 8
    L = lanes(ty)
    tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
    rng = np.random.default_rng(777 + T + waves)
    for n, widths in ((T + 1, np.arange(T + 1)), (1, np.array([T // 2])), (263, rng.integers(0, T + 1, size=263))):
        widths = widths.astype(np.uint8)
        dw = torch.from_numpy(widths).cuda()
        doff, dtotal = fl.widths_to_offsets(ty, dw)
        off = np.concatenate([[0], np.cumsum(widths.astype(np.int64) * 128)]) // esz
        col = values(ty, int(off[-1]), 6100 + n)
        refs = values(ty, n, 6101 + n)
        bases = values(ty, n * L, 6102 + n)
        v = values(ty, n * 1024, 6103 + n)
        blocks = [(b, int(w), col[off[b]:off[b + 1]]) for b, w in enumerate(widths)]
        # FoR decode, references per block and one for all (reference_stride 0)
        got = to_np(fl.unfor_pack_widths(dw, doff, to_dev(col), to_dev(refs)), ty)
        want = np.concatenate([oracle.unfor_pack(ty, w, pk, refs[b]) for b, w, pk in blocks])
        assert np.array_equal(got, want), (ty, n, "unfor_pack_widths")
        if n > 1:
            got = to_np(fl.unfor_pack_widths(dw, doff, to_dev(col), to_dev(refs[:1])), ty)
            want = np.concatenate([oracle.unfor_pack(ty, w, pk, refs[0]) for b, w, pk in blocks])
            assert np.array_equal(got, want), (ty, n, "unfor_pack_widths, one reference")
        # FoR encode: over-wide differences are truncated like for_pack::<W> does (macros.rs:73)
        dpk = torch.zeros(int(off[-1]), dtype=tdt, device="cuda:0")
        fl.for_pack_widths(dw, doff, to_dev(v), to_dev(refs), dpk)
        wantp = np.concatenate([oracle.for_pack(ty, w, v[b * 1024:(b + 1) * 1024], refs[b]) for b, w, _ in blocks]
                               + [np.zeros(0, dtype=TYPES[ty][0])])
        assert np.array_equal(to_np(dpk, ty), wantp), (ty, n, "for_pack_widths")
        # Delta decode (transposed order, as the reference returns it) and straight to original order
        got = to_np(fl.undelta_pack_widths(dw, doff, to_dev(col), to_dev(bases)), ty)
        want = np.concatenate([oracle.undelta_pack(ty, w, pk, bases[b * L:(b + 1) * L]) for b, w, pk in blocks])
        assert np.array_equal(got, want), (ty, n, "undelta_pack_widths")
        got = to_np(fl.undelta_pack_widths(dw, doff, to_dev(col), to_dev(bases), untranspose=True), ty)
        assert np.array_equal(got, oracle.batch("untranspose", ty, None, want)), (ty, n, "undelta_pack_untranspose_widths")
        # Delta encode from original order
        dpk = torch.zeros(int(off[-1]), dtype=tdt, device="cuda:0")
        fl.transpose_delta_pack_widths(dw, doff, to_dev(v), to_dev(bases), dpk)
        dl = oracle.batch("delta", ty, None, oracle.batch("transpose", ty, None, v), aux=bases)
        wantp = np.concatenate([oracle.pack(ty, w, dl[b * 1024:(b + 1) * 1024]) for b, w, _ in blocks] + [np.zeros(0, dtype=TYPES[ty][0])])
        assert np.array_equal(to_np(dpk, ty), wantp), (ty, n, "transpose_delta_pack_widths")
    # the encoder chain: values whose blocks span different ranges -> min / max -> widths -> offsets -> for_pack -> unfor_pack
    n = 150
    span_bits = rng.integers(0, T + 1, size=n)
    lo = values(ty, n, 6200)
    v = np.empty(n * 1024, dtype=TYPES[ty][0])
    raw = values(ty, n * 1024, 6201)
    for b in range(n):
        m = np.array((1 << int(span_bits[b])) - 1, dtype=np.uint64).astype(TYPES[ty][0])
        blk = raw[b * 1024:(b + 1) * 1024] & m
        room = np.array(np.iinfo(TYPES[ty][0]).max, dtype=TYPES[ty][0]) - m     # keep min + span inside the type: no wrap
        v[b * 1024:(b + 1) * 1024] = blk + np.minimum(lo[b], room)
    dv = to_dev(v)
    mins, maxs = fl.BitPacking.block_min_max(dv)
    dw = fl.for_widths(mins, maxs)
    vb = v.reshape(n, 1024)
    want_w = np.array([int(int(vb[b].max()) - int(vb[b].min())).bit_length() for b in range(n)], dtype=np.uint8)
    assert np.array_equal(dw.cpu().numpy(), want_w), (ty, "for_widths")
    doff, dtotal = fl.widths_to_offsets(ty, dw)
    dpk = torch.zeros(int(dtotal.item()) // esz, dtype=tdt, device="cuda:0")
    fl.for_pack_widths(dw, doff, dv, mins, dpk)
    back = fl.unfor_pack_widths(dw, doff, dpk, mins)
    assert np.array_equal(to_np(back, ty), v), (ty, "FoR encoder chain is lossless")
    assert int(dtotal.item()) == int(want_w.astype(np.int64).sum()) * 128
    # for_widths on arbitrary pairs: wrapping difference, bit length
    a, b2 = values(ty, 1000, 6300), values(ty, 1000, 6301)
    gw = fl.for_widths(to_dev(a), to_dev(b2)).cpu().numpy()
    ww = np.array([int((int(y) - int(x)) % (1 << T)).bit_length() for x, y in zip(a, b2)], dtype=np.uint8)
    assert np.array_equal(gw, ww)
    # per-block device checks: the block is skipped, everything else is done, the flag says why
    n = 40
    widths = rng.integers(1, T + 1, size=n).astype(np.uint8)
    off = np.concatenate([[0], np.cumsum(widths.astype(np.int64) * 128)])
    total = int(off[-1])
    col = values(ty, total // esz, 6400)
    refs, bases = values(ty, n, 6401), values(ty, n * L, 6402)
    dw, doff = torch.from_numpy(widths).cuda(), torch.from_numpy(off[:-1].copy()).cuda()
    good_for = to_np(fl.unfor_pack_widths(dw, doff, to_dev(col), to_dev(refs)), ty)
    good_delta = to_np(fl.undelta_pack_widths(dw, doff, to_dev(col), to_dev(bases)), ty)
    bad_w = widths.copy()
    bad_w[7] = T + 1
    cases = [("width", 1, torch.from_numpy(bad_w).cuda(), doff, 7)]
    for what, status, delta_off in (("misaligned", 4, 8), ("outside", 6, 1 << 40)):
        boff = off[:-1].copy()
        boff[5] += delta_off
        cases.append((what, status, dw, torch.from_numpy(boff).cuda(), 5))
    for what, status, w_, o_, skipped in cases:
        for name, call, good in (("unfor_pack_widths", lambda **k: fl.unfor_pack_widths(w_, o_, to_dev(col), to_dev(refs), **k), good_for),
                                 ("undelta_pack_widths", lambda **k: fl.undelta_pack_widths(w_, o_, to_dev(col), to_dev(bases), **k), good_delta)):
            with pytest.raises(fl.FastLanesError) as ei:
                call()
            assert ei.value.status == status, (what, name)
            out = torch.zeros(n * 1024, dtype=tdt, device="cuda:0")
            call(output=out, check=False)
            g = to_np(out, ty)
            keep = np.ones(n * 1024, dtype=bool)
            keep[skipped * 1024:(skipped + 1) * 1024] = False
            assert np.array_equal(g[keep], good[keep]) and not g[~keep].any(), (what, name)
        v = values(ty, n * 1024, 6403)
        for name, call in (("for_pack_widths", lambda o: fl.for_pack_widths(w_, o_, to_dev(v), to_dev(refs), o)),
                           ("transpose_delta_pack_widths", lambda o: fl.transpose_delta_pack_widths(w_, o_, to_dev(v), to_dev(bases), o))):
            guard = torch.full((total // esz,), 0x5A, dtype=tdt, device="cuda:0")
            with pytest.raises(fl.FastLanesError) as ei:
                call(guard)
            assert ei.value.status == status, (what, name)
            lo5, k5 = int(off[skipped]) // esz, packed_len(ty, int(widths[skipped]))
            assert (to_np(guard, ty)[lo5:lo5 + k5] == 0x5A).all(), (what, name, "the skipped block was written")
    # a column of width-0 blocks has no packed bytes: FoR yields the references, Delta the bases repeated down every lane
    z = torch.zeros(3, dtype=torch.uint8, device="cuda:0")
    zoff, _ = fl.widths_to_offsets(ty, z)
    empty = torch.empty(0, dtype=tdt, device="cuda:0")
    r3, b3 = values(ty, 3, 6500), values(ty, 3 * L, 6501)
    assert np.array_equal(to_np(fl.unfor_pack_widths(z, zoff, empty, to_dev(r3)), ty), np.repeat(r3, 1024))
    want = np.concatenate([oracle.undelta_pack(ty, 0, np.zeros(0, dtype=TYPES[ty][0]), b3[b * L:(b + 1) * L]) for b in range(3)])
    assert np.array_equal(to_np(fl.undelta_pack_widths(z, zoff, empty, to_dev(b3)), ty), want)
    fl.for_pack_widths(z, zoff, to_dev(values(ty, 3 * 1024, 1)), to_dev(r3), empty)
    fl.transpose_delta_pack_widths(z, zoff, to_dev(values(ty, 3 * 1024, 1)), to_dev(b3), empty)
    # argument checks of the mirror
    with pytest.raises(ValueError):
        fl.unfor_pack_widths(dw, doff, to_dev(col), to_dev(refs[:3]))
    with pytest.raises(ValueError):
        fl.undelta_pack_widths(dw, doff, to_dev(col), to_dev(bases[:L]))


@pytest.mark.parametrize("ty", TYS)
def test_delta_over_a_batch_of_small_arrays(fl, oracle, ty):
    """fl_<ty>_undelta_pack_batch / _transpose_delta_pack_batch: Delta's fused decode (delta.rs:47-63), its original-order form and the
    fused encode over many small arrays in ONE launch -- every array against the oracle's per-block calls; ragged counts incl. 0,
    widths incl. 0 and T, guard elements behind every output, the device-side checks of the per-array pointers and widths."""
    import torch
    T, L = tbits(ty), lanes(ty)
    tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
    rng = np.random.default_rng(199 + T)
    counts = [64, 1, 0, 5, 64, 33, 7, 2, 130] + [int(x) for x in rng.integers(0, 70, size=30)]
    widths = [int(x) for x in rng.integers(0, T + 1, size=len(counts))]
    widths[0], widths[1], widths[4] = T, 0, 7 % (T + 1)
    packed_np = [values(ty, n * packed_len(ty, w), 9000 + a) for a, (n, w) in enumerate(zip(counts, widths))]
    bases_np = [values(ty, n * L, 9100 + a) for a, n in enumerate(counts)]
    empty = lambda: torch.empty(0, dtype=tdt, device="cuda:0")
    packed = [to_dev(p) if p.size else empty() for p in packed_np]
    bases = [to_dev(b) if b.size else empty() for b in bases_np]
    guard = 0xA5
    for untranspose in (False, True):
        outs = [torch.full((n * 1024 + 64,), guard, dtype=tdt, device="cuda:0") for n in counts]
        fl.Batch(packed, [o[:n * 1024] for o, n in zip(outs, counts)], widths, bases=bases).undelta_pack(untranspose=untranspose, check=True)
        for a, (n, w) in enumerate(zip(counts, widths)):
            got = to_np(outs[a], ty)
            want = oracle.batch("undelta_pack", ty, w, packed_np[a], aux=bases_np[a], n_blocks=n) if n else np.zeros(0, dtype=TYPES[ty][0])
            if untranspose and n:
                want = oracle.batch("untranspose", ty, None, want)
            assert np.array_equal(got[:n * 1024], want), (ty, a, n, w, untranspose)
            assert (got[n * 1024:] == guard).all(), "wrote past the end of an array"
    # encode from original order: pack::<W>(delta(transpose(v), bases)), truncated like pack::<W> does
    vals_np = [values(ty, n * 1024, 9200 + a) for a, n in enumerate(counts)]
    vals = [to_dev(v) if v.size else empty() for v in vals_np]
    pouts = [torch.full((n * packed_len(ty, w) + 64,), guard, dtype=tdt, device="cuda:0") for n, w in zip(counts, widths)]
    fl.Batch([p[:n * packed_len(ty, w)] for p, n, w in zip(pouts, counts, widths)], vals, widths, bases=bases).transpose_delta_pack(check=True)
    for a, (n, w) in enumerate(zip(counts, widths)):
        got = to_np(pouts[a], ty)
        k = n * packed_len(ty, w)
        if k:
            want = oracle.batch("pack", ty, w, oracle.batch("delta", ty, None, oracle.batch("transpose", ty, None, vals_np[a]), aux=bases_np[a]))
            assert np.array_equal(got[:k], want), (ty, a, n, w, "transpose_delta_pack_batch")
        assert (got[k:] == guard).all(), "wrote past the end of a packed array"
    # a width > T arriving in HBM: that array is skipped and flagged, the others are decoded
    b = fl.Batch(packed[:4], [torch.zeros(n * 1024, dtype=tdt, device="cuda:0") for n in counts[:4]], widths[:4], bases=bases[:4])
    b.d_widths[3] = T + 1
    with pytest.raises(fl.FastLanesError) as ei:
        b.undelta_pack(check=True)
    assert ei.value.status == 1
    assert np.array_equal(to_np(b.unpacked[0], ty), oracle.batch("undelta_pack", ty, widths[0], packed_np[0], aux=bases_np[0], n_blocks=counts[0]))
    assert not to_np(b.unpacked[3], ty).any()
    # the mirror's own checks
    with pytest.raises(ValueError):
        fl.Batch(packed[:2], [torch.zeros(n * 1024, dtype=tdt, device="cuda:0") for n in counts[:2]], widths[:2]).undelta_pack()
    with pytest.raises(ValueError):
        fl.Batch(packed[:2], [torch.zeros(n * 1024, dtype=tdt, device="cuda:0") for n in counts[:2]], widths[:2], bases=[bases[0], bases[0][:L - 1]])


@pytest.mark.parametrize("ty", TYS)
def test_batch_of_small_arrays_vs_oracle(fl, oracle, ty):
    """fl_<ty>_unpack_batch / _pack_batch: many small arrays (a columnar engine's chunks), each with its own width and block
    count, given as device arrays of pointers -- ONE launch; every array against the oracle's per-block loop
    (bitpacking.rs:109-129, :76-96).  Ragged counts incl. 0, widths incl. 0 and T, over-wide pack inputs, guard bytes."""
    import torch
    T = tbits(ty)
    esz = T // 8
    tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
    rng = np.random.default_rng(99 + T)
    counts = [64, 1, 0, 5, 64, 33, 7, 64, 2, 130] + [int(x) for x in rng.integers(0, 70, size=40)]
    widths = [int(x) for x in rng.integers(0, T + 1, size=len(counts))]
    widths[0], widths[1], widths[4] = T, 0, 7 % (T + 1)
    packed_np = [values(ty, n * packed_len(ty, w), 7000 + a) for a, (n, w) in enumerate(zip(counts, widths))]
    packed = [to_dev(p) if p.size else torch.empty(0, dtype=tdt, device="cuda:0") for p in packed_np]
    guard = 0xA5 if ty == "u8" else 0xA5A5
    outs = [torch.full((n * 1024 + 64,), guard, dtype=tdt, device="cuda:0") for n in counts]
    batch = fl.Batch(packed, [o[:n * 1024] for o, n in zip(outs, counts)], widths)
    batch.unpack(check=True)
    for a, (n, w) in enumerate(zip(counts, widths)):
        got = to_np(outs[a], ty)
        want = oracle.batch("unpack", ty, w, packed_np[a], n_blocks=n) if n else np.zeros(0, dtype=TYPES[ty][0])
        assert np.array_equal(got[:n * 1024], want), (ty, a, n, w)
        assert (got[n * 1024:] == guard).all(), "wrote past the end of an array"
    # pack back from full-entropy values: truncation exactly like pack::<W> (macros.rs:73; none for W == T, :58)
    vals_np = [values(ty, n * 1024, 8000 + a) for a, n in enumerate(counts)]
    vals = [to_dev(v) if v.size else torch.empty(0, dtype=tdt, device="cuda:0") for v in vals_np]
    pouts = [torch.full((n * packed_len(ty, w) + 64,), guard, dtype=tdt, device="cuda:0") for n, w in zip(counts, widths)]
    fl.Batch([p[:n * packed_len(ty, w)] for p, n, w in zip(pouts, counts, widths)], vals, widths).pack(check=True)
    for a, (n, w) in enumerate(zip(counts, widths)):
        got = to_np(pouts[a], ty)
        k = n * packed_len(ty, w)
        want = oracle.batch("pack", ty, w, vals_np[a]) if k else np.zeros(0, dtype=TYPES[ty][0])
        assert np.array_equal(got[:k], want), (ty, a, n, w)
        assert (got[k:] == guard).all()
    # ... and with FoR's bodies: one reference per array (a chunk's frame of reference), unfor_pack::<W> / for_pack::<W> per block
    refs = [int(x) for x in values(ty, len(counts), 9100)]
    fouts = [torch.full((n * 1024 + 64,), guard, dtype=tdt, device="cuda:0") for n in counts]
    fl.Batch(packed, [o[:n * 1024] for o, n in zip(fouts, counts)], widths, references=refs).unpack(check=True)
    fpouts = [torch.full((n * packed_len(ty, w) + 64,), guard, dtype=tdt, device="cuda:0") for n, w in zip(counts, widths)]
    fl.Batch([p[:n * packed_len(ty, w)] for p, n, w in zip(fpouts, counts, widths)], vals, widths, references=refs).pack(check=True)
    for a, (n, w) in enumerate(zip(counts, widths)):
        if n == 0:
            continue
        r = np.full(n, refs[a], dtype=TYPES[ty][0])
        assert np.array_equal(to_np(fouts[a], ty)[:n * 1024], oracle.batch("unfor_pack", ty, w, packed_np[a], aux=r, n_blocks=n)), (ty, a, "unfor")
        k = n * packed_len(ty, w)
        if k:
            assert np.array_equal(to_np(fpouts[a], ty)[:k], oracle.batch("for_pack", ty, w, vals_np[a], aux=r)), (ty, a, "for")
        assert (to_np(fouts[a], ty)[n * 1024:] == guard).all() and (to_np(fpouts[a], ty)[k:] == guard).all()
    # errors: a width > T is refused on the host side of the mirror, and flagged by the kernel when it arrives in HBM
    with pytest.raises(fl.FastLanesError):
        fl.Batch(packed[:1], [outs[0][:counts[0] * 1024]], [T + 1])
    b2 = fl.Batch(packed[:2], [outs[0][:counts[0] * 1024], outs[1][:counts[1] * 1024]], widths[:2])
    b2.d_widths[0] = T + 1
    with pytest.raises(fl.FastLanesError) as ei:
        b2.unpack(check=True)
    assert ei.value.status == 1
    with pytest.raises(ValueError):
        fl.Batch(packed[:1], [outs[0][:1024 * (counts[0] - 1)]], widths[:1])
    # a bound smaller than an array's block count would leave its tail undecoded: the kernel flags it
    b3 = fl.Batch(packed[:1], [outs[0][:counts[0] * 1024]], widths[:1])
    b3.max_blocks = counts[0] - 8
    with pytest.raises(fl.FastLanesError) as ei:
        b3.unpack(check=True)
    assert ei.value.status == 6


def test_mixed_width_fuzz_shapes_and_occupancies(fl, oracle, kernel_policy):
    """Seeded fuzz of the device-resident mixed-width kernels over (type, block count, widths incl. 0 and T, waves per SIMD,
    blocks per wavefront, prefetch): unpack_widths / pack_widths / unpack_single_widths against the oracle's per-block loop."""
    import torch
    rng = np.random.default_rng(int(os.environ.get("FL_FUZZ_SEED", "2025")))
    for _ in range(int(os.environ.get("FL_FUZZ_ITERS", "48"))):
        ty = TYS[rng.integers(0, 4)]
        T = tbits(ty)
        esz = T // 8
        tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 65, 127, 129, 300]))
        waves = int(rng.choice([0, 3, 4, 6, 8]))
        bpw = int(rng.choice([0, 1, 2, 3, 8]))
        prefetch = int(rng.integers(0, 2)) if bpw >= 2 else 0      # all of a wavefront's blocks requested up front by LDS-DMA
        kernel_policy(2 + 256 * waves + 65536 * bpw + (prefetch << 24))
        widths = rng.integers(0, T + 1, size=n).astype(np.uint8)
        off = np.concatenate([[0], np.cumsum(widths.astype(np.int64) * 128)])
        seed = int(rng.integers(0, 1 << 30))
        col = values(ty, int(off[-1]) // esz, seed)
        v = values(ty, n * 1024, seed + 1)
        dw = torch.from_numpy(widths).cuda()
        doff, dtotal = fl.widths_to_offsets(ty, dw)
        assert int(dtotal.item()) == off[-1]
        want = np.concatenate([oracle.unpack(ty, int(w), col[off[b] // esz:off[b + 1] // esz]) for b, w in enumerate(widths)])
        dcol = to_dev(col)
        assert np.array_equal(to_np(fl.unpack_widths(dw, doff, dcol), ty), want), (ty, n, waves, bpw, seed, "unpack")
        dpk = torch.zeros(int(off[-1]) // esz, dtype=tdt, device="cuda:0")
        fl.pack_widths(dw, doff, to_dev(v), dpk)
        wantp = np.concatenate([oracle.pack(ty, int(w), v[b * 1024:(b + 1) * 1024]) for b, w in enumerate(widths)]
                               + [np.zeros(0, dtype=TYPES[ty][0])])
        assert np.array_equal(to_np(dpk, ty), wantp), (ty, n, waves, bpw, seed, "pack")
        idx = rng.integers(0, n * 1024, size=257).astype(np.int64)
        got = to_np(fl.unpack_single_widths(dw, doff, dcol, torch.from_numpy(idx).cuda()), ty)
        assert np.array_equal(got, want[idx]), (ty, n, seed, "unpack_single")


def test_widths_to_offsets_large(fl):
    """The three-launch device scan across many 4096-block chunks, ragged tail, against numpy."""
    import torch
    for n in (4095, 4096, 4097, 3 * 4096 + 17, 1_000_003):
        w = np.random.default_rng(n).integers(0, 33, size=n).astype(np.uint8)
        off, total = fl.widths_to_offsets("u32", torch.from_numpy(w).cuda())
        want = np.concatenate([[0], np.cumsum(w.astype(np.int64) * 128)])
        assert np.array_equal(off.cpu().numpy(), want[:-1]), n
        assert int(total.item()) == want[-1]


def test_second_gpu_if_present(fl, oracle):
    """Multi-GPU path on real devices: a rank's slice decoded on cuda:1 (device switch of the Python
    mirror, a plan bound to device 1, device-resident widths on device 1).  Skipped on 1-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    from fastlanes_amd.sharding import shard_mixed
    dev = torch.device("cuda:1")
    n = 1000
    widths = (1 + np.arange(n) % 32).astype(np.uint8)
    s, c, b0, nb = shard_mixed(widths, 2, 1)
    off = np.concatenate([[0], np.cumsum(widths.astype(np.int64) * 128)])
    col = values("u32", int(off[-1]) // 4, 77)
    sl = torch.from_numpy(col[b0 // 4:(b0 + nb) // 4].view(np.uint8)).to(dev).view(torch.uint32)
    want = np.concatenate([oracle.unpack("u32", int(w), col[off[b] // 4:off[b + 1] // 4]) for b, w in enumerate(widths)])[s * 1024:]
    plan = fl.MixedWidthPlan("u32", widths[s:s + c], device=dev)
    out = plan.unpack(sl)
    assert out.device == dev
    assert np.array_equal(out.view(torch.uint8).cpu().numpy().view(np.uint32), want)
    dw = torch.from_numpy(widths[s:s + c]).to(dev)
    doff, _ = fl.widths_to_offsets("u32", dw)
    out2 = fl.unpack_widths(dw, doff, sl)
    assert torch.equal(out2.view(torch.int32), out.view(torch.int32))
    pk7 = values("u32", 50 * 224, 78)
    got = fl.BitPacking.unpack(7, torch.from_numpy(pk7.view(np.uint8)).to(dev).view(torch.uint32))   # current device stays 0
    assert got.device == dev and torch.cuda.current_device() == 0
    assert np.array_equal(got.view(torch.uint8).cpu().numpy().view(np.uint32), oracle.batch("unpack", "u32", 7, pk7))
    with pytest.raises(ValueError):                                       # tensors of two devices in one call
        fl.BitPacking.unpack(7, to_dev(pk7), output=torch.empty(50 * 1024, dtype=torch.uint32, device=dev))
    plan.close()


def test_config5_u32_mixed_widths_10B_integers(fl, oracle):
    """u32, width[b] = 1 + b % 32, 9 765 625 blocks (10 B integers): the per-GPU slices of the
    8-way sharding are exercised on one GPU one after the other; pack(unpack(x)) == x and
    sampled blocks (first/last of every slice) match the oracle."""
    import torch
    from fastlanes_amd.sharding import shard_mixed
    n = 9_765_625
    widths = (1 + np.arange(n) % 32).astype(np.uint8)
    total_bytes = int(widths.astype(np.int64).sum()) * 128
    assert total_bytes == 20_624_988_800
    col = _rand_dev(total_bytes, 46).view(torch.uint32)
    plan = fl.MixedWidthPlan("u32", widths)
    out = plan.unpack(col)
    assert torch.equal(plan.pack(out).view(torch.int32), col.view(torch.int32))
    samples = set()
    for r in range(8):
        s, c, b0, nb = shard_mixed(widths, 8, r)
        samples |= {s, s + c - 1}
        # a rank's slice decoded on its own (own plan over its widths) equals the same rows of the whole
        if r in (0, 7):
            sub = fl.MixedWidthPlan("u32", widths[s:s + c])
            o2 = sub.unpack(col[b0 // 4:(b0 + nb) // 4])
            assert torch.equal(o2.view(torch.int32), out[s * 1024:(s + c) * 1024].view(torch.int32))
            sub.close()
    off = np.concatenate([[0], np.cumsum(widths.astype(np.int64) * 32)])
    for b in sorted(samples):
        w = int(widths[b])
        pk = to_np(col[off[b]:off[b] + 32 * w], "u32")
        assert np.array_equal(to_np(out[b * 1024:(b + 1) * 1024], "u32"), oracle.unpack("u32", w, pk)), b
    plan.close()


# ---------------------------------------------------------------------------
# extensions: fused decode to / encode from the ORIGINAL order.
# Defined as compositions of reference functions, so the oracle composition is the spec.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("policy", [0, 1, 2])
@pytest.mark.parametrize("ty", TYS)
def test_fused_transpose_extensions_vs_oracle_composition(fl, oracle, kernel_policy, ty, policy):
    kernel_policy(policy)
    import torch
    T = tbits(ty)
    n = 37
    for w in range(T + 1):
        seed = 11000 + 64 * T + w
        pk = values(ty, n * packed_len(ty, w), seed)
        bases = values(ty, n * lanes(ty), seed + 1)
        v = values(ty, n * 1024, seed + 2)
        # decode: untranspose(undelta_pack(pk, bases))
        got = to_np(fl.Delta.undelta_pack_untranspose(w, to_dev(pk), to_dev(bases)), ty)
        want = oracle.batch("untranspose", ty, None, oracle.batch("undelta_pack", ty, w, pk, aux=bases, n_blocks=n))
        assert np.array_equal(got, want), (ty, w, "undelta_pack_untranspose")
        # encode: pack(delta(transpose(v), bases))
        got = to_np(fl.Delta.transpose_delta_pack(w, to_dev(v), to_dev(bases)), ty)
        want = oracle.batch("pack", ty, w, oracle.batch("delta", ty, None, oracle.batch("transpose", ty, None, v), aux=bases))
        assert np.array_equal(got, want), (ty, w, "transpose_delta_pack")
    # full-width round trip: decode(encode(v)) == v in the original order
    v = values(ty, n * 1024, 5)
    bases = values(ty, n * lanes(ty), 6)
    enc = fl.Delta.transpose_delta_pack(T, to_dev(v), to_dev(bases))
    assert np.array_equal(to_np(fl.Delta.undelta_pack_untranspose(T, enc, to_dev(bases)), ty), v)


@pytest.mark.parametrize("ty", TYS)
def test_fused_consumers_vs_oracle(fl, oracle, ty):
    """unpack_block_sums / block_min_max (extensions): reductions of the oracle's outputs."""
    T = tbits(ty)
    n = 37
    for w in range(T + 1):
        pk = values(ty, n * packed_len(ty, w), 13000 + 64 * T + w)
        got = fl.BitPacking.unpack_block_sums(w, to_dev(pk), n_blocks=n).cpu().numpy().view(np.uint64)
        un = oracle.batch("unpack", ty, w, pk, n_blocks=n).reshape(n, 1024)
        with np.errstate(over="ignore"):
            want = un.astype(np.uint64).sum(axis=1, dtype=np.uint64)
        assert np.array_equal(got, want), (ty, w)
    v = values(ty, n * 1024, 77 + T)
    v[5 * 1024:6 * 1024] = v[5 * 1024]          # a constant block
    v[7 * 1024 + 1023] = np.iinfo(TYPES[ty][0]).max
    v[9 * 1024] = 0
    mins, maxs = fl.BitPacking.block_min_max(to_dev(v))
    assert np.array_equal(to_np(mins, ty), v.reshape(n, 1024).min(axis=1))
    assert np.array_equal(to_np(maxs, ty), v.reshape(n, 1024).max(axis=1))
    # the encoder loop these feed: FoR reference = min, width = bits(max - min)
    ref = to_np(mins, ty)
    span = (to_np(maxs, ty) - ref).astype(np.uint64)
    w = int(max(int(x).bit_length() for x in span))
    pk = fl.FoR.for_pack(w, to_dev(v), mins)
    assert np.array_equal(to_np(fl.FoR.unfor_pack(w, pk, mins), ty), v)


def test_functor_api_user_kernel_dict_decode(fl, oracle):
    """A user-written fused kernel (examples/fused_dict_decode.hip) built on the device functor
    API: out[idx] = dict[code] spliced into the unpack loop, vs dict[oracle.unpack(..)]."""
    import ctypes
    import torch
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_examples()["fused_dict_decode"])
    lib.example_dict_unpack_u32_w8.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    n = 70
    pk = values("u32", n * 256, 21)
    dic = values("u32", 256, 22)
    dpk, ddic = to_dev(pk), to_dev(dic)
    out = torch.empty(n * 1024, dtype=torch.uint32, device="cuda:0")
    rc = lib.example_dict_unpack_u32_w8(dpk.data_ptr(), ddic.data_ptr(), out.data_ptr(), n,
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    want = dic[oracle.batch("unpack", "u32", 8, pk)]
    assert np.array_equal(to_np(out, "u32"), want)


def test_plain_c_caller_of_the_c_abi(fl):
    """examples/column_decode.c: a C99 program (no C++, no Python) that encodes, decodes and point-reads a mixed-width column
    through the C ABI exactly as a cgo / JNI / Rust-FFI binding would -- device tier, host tier and the device prefix sum."""
    import subprocess
    import __graft_entry__ as ge
    r = subprocess.run([ge.build_examples()["column_decode"]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr)


def test_multi_device_driver_through_the_c_abi(fl):
    """examples/multi_gpu_decode.c: one process, one host thread per visible device (hipSetDevice + its own stream), no
    torch, no RCCL -- the weak-scaled config-2 column on every device and the strong-scaled config-5 column split by
    contiguous block range, each slice's first / last / sampled blocks verified against a scalar decode.
    Runs on however many devices exist; --replicas 2 drives the N-thread path on a 1-GPU box as well."""
    import subprocess
    import torch
    import __graft_entry__ as ge
    exe = ge.build_examples()["multi_gpu_decode"]
    ndev = torch.cuda.device_count()
    for extra, threads in ((["--replicas", "1"], ndev), (["--replicas", "2"], 2 * ndev)):
        r = subprocess.run([exe, "--blocks", "300000", "--strong-blocks", "600001", "--steps", "3", "--warmup", "1"] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
        d = json.loads(r.stdout)
        assert d["devices"] == ndev and d["threads"] == threads and d["correct"] is True and d["collective"] == "none"
        weak, strong = d["weak_u32_w7_unpack"], d["strong_u32_mixed_unpack"]
        assert all(t["correct"] and t["GBps"] > 0 for t in weak["per_thread"]) and weak["Gint_per_s"] > 0
        sl = strong["per_thread"]
        assert sum(t["blocks"] for t in sl) == 600001 and sl[0]["first_block"] == 0
        assert all(sl[i]["first_block"] + sl[i]["blocks"] == sl[i + 1]["first_block"] for i in range(len(sl) - 1))
        assert all(t["correct"] and t["GBps"] > 0 for t in sl) and strong["Gint_per_s"] > 0
        assert sorted({t["device"] for t in sl}) == list(range(ndev))


def test_fill_random_is_the_counter_based_stream(fl):
    """fl_fill_random writes splitmix64 of the global word index: identical to tests/datagen.py, so a host
    can regenerate any part of a device-resident column."""
    import torch
    from datagen import splitmix64
    lib = fl.load()
    for n_words, seed in ((1, 0), (1000, 42), (65536 * 256 + 77, 1234)):
        t = torch.zeros(n_words + 2, dtype=torch.int64, device="cuda:0")
        assert lib.fl_fill_random(t.data_ptr() + 8, n_words * 8, seed, None) == 0
        torch.cuda.synchronize()
        got = t.cpu().numpy().view(np.uint64)
        assert got[0] == 0 and got[-1] == 0                     # nothing outside the range
        assert np.array_equal(got[1:-1], splitmix64(n_words, seed))


def test_functor_api_user_kernel_on_iterate_rows(fl, oracle):
    """examples/iterate_running_max.hip: a stateful body spliced into fl::iterate_rows (the iterate! counterpart,
    macros.rs:11-32) -- a running maximum along every FastLanes lane in row order, the shape of Delta::undelta
    (delta.rs:36-45) -- against the same walk written with the oracle's index(row, lane) (macros.rs:20-24)."""
    import ctypes
    import torch
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_examples()["iterate_running_max"])
    lib.example_running_max_u32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    n = 45
    v = values("u32", n * 1024, 31)
    base = values("u32", n * 32, 32) >> np.uint32(1)       # about half of the lanes start above their first values
    out = torch.empty(n * 1024, dtype=torch.uint32, device="cuda:0")
    dv, dbase = to_dev(v), to_dev(base)                    # keep the device tensors alive across the raw-pointer call
    rc = lib.example_running_max_u32(dv.data_ptr(), dbase.data_ptr(), out.data_ptr(), n,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    idx = np.array([[oracle.index(r, l) for l in range(32)] for r in range(32)])      # [row][lane]
    want = np.empty_like(v)
    for b in range(n):
        blk = v[b * 1024:(b + 1) * 1024]
        run = base[b * 32:(b + 1) * 32].copy()
        for r in range(32):
            run = np.maximum(run, blk[idx[r]])
            want[b * 1024 + idx[r]] = run
    assert np.array_equal(to_np(out, "u32"), want)
    # and the library's own user of iterate_rows: Delta::undelta == the same walk with a running sum
    got = to_np(fl.Delta.undelta(to_dev(v), to_dev(base)), "u32")
    wsum = np.empty_like(v)
    for b in range(n):
        run = base[b * 32:(b + 1) * 32].copy()
        for r in range(32):
            run = run + v[b * 1024 + idx[r]]
            wsum[b * 1024 + idx[r]] = run
    assert np.array_equal(got, wsum)


@pytest.mark.parametrize("ty", TYS)
def test_results_do_not_depend_on_the_tile_map(fl, oracle, ty):
    """fl_kernels.hpp: xcd_tile -- which workgroup takes which tile is chosen for speed (round 4: read-dominated kernels walk the
    column in windows) and must never change a byte.  Every family under windows of 2^8, 2^9 and 2^12 blocks (many windows, a
    short last one: the block count is ragged against all of them) and under the whole-column map, with both kernel designs,
    against the default's output -- which the other tests tie to the oracle -- and, for unpack, against the oracle directly."""
    import torch
    T, L = tbits(ty), lanes(ty)
    n = 5000 + 37
    w = {8: 5, 16: 9, 32: 7, 64: 17}[T]
    lib = fl.load()
    v = values(ty, n * 1024, 4100 + T)
    pk = values(ty, n * packed_len(ty, w), 4200 + T)
    bases = values(ty, n * L, 4300 + T)
    refs = values(ty, n, 4400 + T)
    dv, dpk, db, dr = to_dev(v), to_dev(pk), to_dev(bases), to_dev(refs)
    widths = torch.from_numpy((np.arange(n) * 7 % (T + 1)).astype(np.uint8)).cuda()
    offsets, total = fl.widths_to_offsets(ty, widths)
    col = to_dev(values(ty, max(int(total.item()) // (T // 8), 16), 4500 + T))
    chunks = [dpk[a * 100 * packed_len(ty, w):(a + 1) * 100 * packed_len(ty, w)] for a in range(50)]

    def everything():
        outs = [torch.empty(100 * 1024, dtype=dv.dtype, device="cuda:0") for _ in range(50)]
        fl.Batch(chunks, outs, [w] * 50).unpack(check=True)
        r = [fl.BitPacking.unpack(w, dpk), fl.BitPacking.pack(w, dv), fl.FoR.unfor_pack(w, dpk, dr), fl.FoR.for_pack(w, dv, dr),
             fl.Delta.delta(dv, db), fl.Delta.undelta(dv, db), fl.Delta.undelta_pack(w, dpk, db), fl.Transpose.transpose(dv),
             fl.Transpose.untranspose(dv), fl.Delta.undelta_pack_untranspose(w, dpk, db), fl.Delta.transpose_delta_pack(w, dv, db),
             fl.BitPacking.unpack_compare(w, dpk, "<=", (1 << w) // 3), fl.BitPacking.unpack_block_sums(w, dpk),
             *fl.BitPacking.block_min_max(dv), fl.unpack_widths(widths, offsets, col), torch.cat(outs),
             fl.unfor_pack_widths(widths, offsets, col, dr), fl.undelta_pack_widths(widths, offsets, col, db),
             fl.undelta_pack_widths(widths, offsets, col, db, untranspose=True),
             fl.for_pack_widths(widths, offsets, dv, dr, torch.zeros_like(col)),
             fl.transpose_delta_pack_widths(widths, offsets, dv, db, torch.zeros_like(col))]
        torch.cuda.synchronize()
        return [t.view(torch.uint8).clone() for t in r]

    try:
        want = everything()
        assert np.array_equal(want[0].cpu().numpy().view(TYPES[ty][0]), oracle.batch("unpack", ty, w, pk, n_blocks=n))
        for design in (0, 1, 2):
            for window in (8, 9, 12, 31):
                lib.fl_internal_set_kernel_policy(design + (window << 25))
                assert lib.fl_internal_get_kernel_policy() == design + (window << 25)
                got = everything()
                for i, (g, e) in enumerate(zip(got, want)):
                    assert torch.equal(g, e), (ty, design, window, i)
    finally:
        lib.fl_internal_set_kernel_policy(0)


def test_fl_check_device_refuses_foreign_pointers():
    """FL_CHECK_DEVICE=1 (include/fastlanes_amd.h "Threading and device selection"): every device-tier entry verifies that its
    pointers are memory the current device can use and that the stream is one of its streams, and returns FL_ERR_DEVICE (7)
    instead of launching.  The variable is read once per process, hence a child process.  Without it the same call with a host
    pointer is undefined behaviour, exactly like a raw kernel launch -- not exercised."""
    import subprocess
    import sys
    code = r'''
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import fastlanes_amd as fl
lib = fl.load()
n = 8
dev_pk = torch.zeros(n * 224, dtype=torch.int32, device="cuda:0")
dev_out = torch.zeros(n * 1024, dtype=torch.int32, device="cuda:0")
host = np.zeros(n * 1024, dtype=np.uint32)
pinned = torch.zeros(n * 1024, dtype=torch.int32).pin_memory()
st = torch.cuda.Stream()
res = {
    "device_ok": lib.fl_u32_unpack(7, dev_pk.data_ptr(), dev_out.data_ptr(), n, None),
    "stream_ok": lib.fl_u32_unpack(7, dev_pk.data_ptr(), dev_out.data_ptr(), n, ctypes.c_void_p(st.cuda_stream)),
    "host_out": lib.fl_u32_unpack(7, dev_pk.data_ptr(), host.ctypes.data, n, None),
    "host_in": lib.fl_u32_pack(7, host.ctypes.data, dev_pk.data_ptr(), n, None),
    "pinned_out": lib.fl_u32_unpack(7, dev_pk.data_ptr(), pinned.data_ptr(), n, None),
    "fill_host": lib.fl_fill_random(host.ctypes.data, 4096, 1, None),
    "null_is_not_a_device_error": lib.fl_u32_unpack(7, None, dev_out.data_ptr(), n, None),
    "host_widths": lib.fl_u32_unpack_widths(host.ctypes.data, dev_out.data_ptr(), dev_pk.data_ptr(), 896 * n, dev_out.data_ptr(), n, None, None),
}
torch.cuda.synchronize()
print(res)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=dict(os.environ, FL_CHECK_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    res = eval([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res == {"device_ok": 0, "stream_ok": 0, "host_out": 7, "host_in": 7, "pinned_out": 0, "fill_host": 7,
                   "null_is_not_a_device_error": 3, "host_widths": 7}, res


def test_functor_api_user_kernel_on_pack_rows(fl, oracle):
    """examples/fused_for_pack.hip: a body spliced into fl::pack_rows (the pack! counterpart, macros.rs:34-98) -- frame-of-reference
    encoding with the reference (the block's minimum) computed in the same kernel.  Specified as the oracle composition
    for_pack::<W>(v, min(v)) (ffor.rs:24-36), every width, ragged block count; and the library's unfor_pack with those
    minima brings the values back wherever W covers their range."""
    import ctypes
    import torch
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_examples()["fused_for_pack"])
    lib.example_min_for_pack_u32.argtypes = [ctypes.c_uint] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    n = 45
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for w in range(33):
        # values spread over a window of 2^w above a per-block floor, so that W bits hold value - min exactly
        floor = values("u32", n, 700 + w) >> np.uint32(1)
        v = (np.repeat(floor, 1024) + values("u32", n * 1024, 800 + w, bits=max(w - 1, 0)) if w else np.repeat(floor, 1024)).astype(np.uint32)
        dv = to_dev(v)
        pk = torch.zeros(max(n * 32 * w, 4), dtype=torch.uint32, device="cuda:0")
        mins = torch.zeros(n, dtype=torch.uint32, device="cuda:0")
        assert lib.example_min_for_pack_u32(w, dv.data_ptr(), pk.data_ptr(), mins.data_ptr(), n, st) == 0
        torch.cuda.synchronize()
        want_min = v.reshape(n, 1024).min(axis=1)
        assert np.array_equal(to_np(mins, "u32"), want_min), w
        assert np.array_equal(to_np(pk, "u32")[:n * 32 * w], oracle.batch("for_pack", "u32", w, v, aux=want_min)), w
        if w:
            back = fl.FoR.unfor_pack(w, pk[:n * 32 * w], mins)
            assert np.array_equal(to_np(back, "u32"), v), w
    assert lib.example_min_for_pack_u32(33, dv.data_ptr(), pk.data_ptr(), mins.data_ptr(), n, st) == 1      # width > T


# ---------------------------------------------------------------------------
# several tiles per XCD slot (n_blocks > 256) for every kernel family, and launch plumbing
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("ty", TYS)
def test_many_tiles_all_families(fl, oracle, ty):
    T = tbits(ty)
    n = 1000 + T  # 32+ tiles -> tiles_per_xcd = 4..5, ragged last tile
    v = values(ty, n * 1024, 31 + T)
    refs = values(ty, n, 32 + T)
    bases = values(ty, n * lanes(ty), 33 + T)
    dv, drefs, dbases = to_dev(v), to_dev(refs), to_dev(bases)
    for w in sorted({1, T // 2 + 1, T - 1}):
        pk = values(ty, n * packed_len(ty, w), 34 + T + w)
        dpk = to_dev(pk)
        assert np.array_equal(to_np(fl.BitPacking.pack(w, dv), ty), oracle.batch("pack", ty, w, v))
        assert np.array_equal(to_np(fl.BitPacking.unpack(w, dpk), ty), oracle.batch("unpack", ty, w, pk))
        assert np.array_equal(to_np(fl.FoR.for_pack(w, dv, drefs), ty), oracle.batch("for_pack", ty, w, v, aux=refs))
        assert np.array_equal(to_np(fl.FoR.unfor_pack(w, dpk, drefs), ty), oracle.batch("unfor_pack", ty, w, pk, aux=refs))
        ud = oracle.batch("undelta_pack", ty, w, pk, aux=bases)
        assert np.array_equal(to_np(fl.Delta.undelta_pack(w, dpk, dbases), ty), ud)
        assert np.array_equal(to_np(fl.Delta.undelta_pack_untranspose(w, dpk, dbases), ty),
                              oracle.batch("untranspose", ty, None, ud))
        with np.errstate(over="ignore"):
            sums = oracle.batch("unpack", ty, w, pk).reshape(n, 1024).astype(np.uint64).sum(axis=1, dtype=np.uint64)
        assert np.array_equal(fl.BitPacking.unpack_block_sums(w, dpk).cpu().numpy().view(np.uint64), sums)
    assert np.array_equal(to_np(fl.Delta.delta(dv, dbases), ty), oracle.batch("delta", ty, None, v, aux=bases))
    assert np.array_equal(to_np(fl.Delta.undelta(dv, dbases), ty), oracle.batch("undelta", ty, None, v, aux=bases))
    assert np.array_equal(to_np(fl.Transpose.transpose(dv), ty), oracle.batch("transpose", ty, None, v))
    assert np.array_equal(to_np(fl.Transpose.untranspose(dv), ty), oracle.batch("untranspose", ty, None, v))
    mins, maxs = fl.BitPacking.block_min_max(dv)
    assert np.array_equal(to_np(mins, ty), v.reshape(n, 1024).min(axis=1))
    assert np.array_equal(to_np(maxs, ty), v.reshape(n, 1024).max(axis=1))


def test_streams_and_graph_capture(fl, oracle):
    """Device-tier calls are asynchronous on the caller's stream, allocate nothing and never
    synchronise, so they run on side streams and capture into a HIP graph (replayed twice)."""
    import torch
    n = 300
    pk = values("u32", n * 224, 55)
    dpk = to_dev(pk)
    want = oracle.batch("unpack", "u32", 7, pk)
    out = torch.zeros(n * 1024, dtype=torch.uint32, device="cuda:0")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fl.BitPacking.unpack(7, dpk, output=out)
    side.synchronize()
    assert np.array_equal(to_np(out, "u32"), want)
    widths = (1 + np.arange(n) % 32).astype(np.uint8)
    plan = fl.MixedWidthPlan("u32", widths)
    col = values("u32", plan.packed_bytes // 4, 56)
    dcol = to_dev(col)
    mixed_out = torch.zeros(n * 1024, dtype=torch.uint32, device="cuda:0")
    out.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fl.BitPacking.unpack(7, dpk, output=out)
        plan.unpack(dcol, output=mixed_out)          # 32 launches, one per width
    for _ in range(2):
        out.zero_(); mixed_out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(to_np(out, "u32"), want)
        pos, wantm = 0, []
        for w in widths:
            wantm.append(oracle.unpack("u32", int(w), col[pos:pos + 32 * int(w)]))
            pos += 32 * int(w)
        assert np.array_equal(to_np(mixed_out, "u32"), np.concatenate(wantm))
    plan.close()


def test_mixed_plan_column_beyond_4GiB(fl, oracle):
    """A 4.9 GB unpacked column (600 001 u64 blocks; two width-2 blocks at its ends among width-1 blocks):
    64-bit block addressing -- nothing in the mixed path may assume a 32-bit window."""
    import torch
    n = 600_001
    widths = np.ones(n, dtype=np.uint8)
    widths[0] = 2
    widths[n - 1] = 2
    plan = fl.MixedWidthPlan("u64", widths)
    col = _rand_dev(plan.packed_bytes, 47).view(torch.uint64)
    out = plan.unpack(col)
    assert torch.equal(plan.pack(out).view(torch.int64), col.view(torch.int64))
    off = np.concatenate([[0], np.cumsum(widths.astype(np.int64) * 16)])
    for b in (0, 1, 31, 32, n // 2, n - 2, n - 1):
        w = int(widths[b])
        pk = to_np(col[off[b]:off[b] + 16 * w], "u64")
        assert np.array_equal(to_np(out[b * 1024:(b + 1) * 1024], "u64"), oracle.unpack("u64", w, pk)), b
    plan.close()


@pytest.mark.parametrize("ty", TYS)
def test_no_write_past_the_end_of_the_column(fl, ty):
    """Partial wavefronts rely on the buffer descriptor's bounds check to drop the stores of
    blocks past the end: a guard region right behind every output must stay untouched."""
    import torch
    T = tbits(ty)
    tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
    esz = T // 8
    GUARD = 64 * 1024  # bytes

    def guarded(n_elems):
        big = torch.full((n_elems * esz + GUARD,), 0xA5, dtype=torch.uint8, device="cuda:0")
        return big, big[:n_elems * esz].view(tdt)

    for n in (1, 5, 9, 37, 263):
        w = max(1, T // 3)
        pk = to_dev(values(ty, n * packed_len(ty, w), 600 + n))
        v = to_dev(values(ty, n * 1024, 601 + n))
        bases = to_dev(values(ty, n * lanes(ty), 602 + n))
        refs = to_dev(values(ty, n, 603 + n))
        plan = fl.MixedWidthPlan(ty, np.full(n, w, dtype=np.uint8))
        calls = [
            (n * 1024, lambda o: fl.BitPacking.unpack(w, pk, output=o)),
            (n * 1024, lambda o: fl.FoR.unfor_pack(w, pk, refs, output=o)),
            (n * 1024, lambda o: fl.Delta.undelta_pack(w, pk, bases, output=o)),
            (n * 1024, lambda o: fl.Delta.undelta_pack_untranspose(w, pk, bases, output=o)),
            (n * 1024, lambda o: fl.Delta.delta(v, bases, output=o)),
            (n * 1024, lambda o: fl.Delta.undelta(v, bases, output=o)),
            (n * 1024, lambda o: fl.Transpose.transpose(v, output=o)),
            (n * 1024, lambda o: fl.Transpose.untranspose(v, output=o)),
            (n * packed_len(ty, w), lambda o: fl.BitPacking.pack(w, v, output=o)),
            (n * packed_len(ty, w), lambda o: fl.FoR.for_pack(w, v, refs, output=o)),
            (n * packed_len(ty, w), lambda o: fl.Delta.transpose_delta_pack(w, v, bases, output=o)),
            (n * 1024, lambda o: plan.unpack(pk, output=o)),
            (n * packed_len(ty, w), lambda o: plan.pack(v, output=o)),
        ]
        for k, (n_elems, call) in enumerate(calls):
            big, out = guarded(n_elems)
            call(out)
            torch.cuda.synchronize()
            assert bool((big[n_elems * esz:] == 0xA5).all()), (ty, n, k)
        plan.close()


@pytest.mark.parametrize("policy", [0, 1, 2])
def test_random_shapes_fuzz(fl, oracle, kernel_policy, policy):
    """Seeded fuzz over (type, width, op, block count): tail handling of tiles / wavefronts, under the automatic kernel
    choice and with each kernel design forced (policy 2 also draws a random occupancy)."""
    rng = np.random.default_rng(int(os.environ.get("FL_FUZZ_SEED", "2024")) + policy)
    ops = ["pack", "unpack", "for_pack", "unfor_pack", "undelta_pack", "delta", "undelta", "transpose", "untranspose"]
    for _ in range(int(os.environ.get("FL_FUZZ_ITERS", "60"))):
        ty = TYS[rng.integers(0, 4)]
        T = tbits(ty)
        w = int(rng.integers(0, T + 1))
        n = int(rng.choice([1, 2, 7, 8, 9, 31, 32, 33, 63, 64, 65, 255, 256, 257, 300, 511, 777]))
        op = ops[rng.integers(0, len(ops))]
        seed = int(rng.integers(0, 1 << 30))
        kernel_policy(policy if policy != 2 else 2 + 256 * int(rng.choice([0, 3, 4, 5, 6, 8])))
        v = values(ty, n * 1024, seed)
        pk = values(ty, n * packed_len(ty, w), seed + 1)
        refs = values(ty, n, seed + 2)
        bases = values(ty, n * lanes(ty), seed + 3)
        if op == "pack":
            got, want = fl.BitPacking.pack(w, to_dev(v)), oracle.batch("pack", ty, w, v)
        elif op == "unpack":
            got, want = fl.BitPacking.unpack(w, to_dev(pk), n_blocks=n), oracle.batch("unpack", ty, w, pk, n_blocks=n)
        elif op == "for_pack":
            got, want = fl.FoR.for_pack(w, to_dev(v), to_dev(refs)), oracle.batch("for_pack", ty, w, v, aux=refs)
        elif op == "unfor_pack":
            got = fl.FoR.unfor_pack(w, to_dev(pk), to_dev(refs), n_blocks=n)
            want = oracle.batch("unfor_pack", ty, w, pk, aux=refs, n_blocks=n)
        elif op == "undelta_pack":
            got = fl.Delta.undelta_pack(w, to_dev(pk), to_dev(bases))
            want = oracle.batch("undelta_pack", ty, w, pk, aux=bases, n_blocks=n)
        elif op in ("delta", "undelta"):
            got = getattr(fl.Delta, op)(to_dev(v), to_dev(bases))
            want = oracle.batch(op, ty, None, v, aux=bases)
        else:
            got = getattr(fl.Transpose, op)(to_dev(v))
            want = oracle.batch(op, ty, None, v)
        assert np.array_equal(to_np(got, ty), want), (ty, w, op, n, seed)


def test_unpack_compare_u32_u64_every_width(fl, oracle):
    """unpack_compare of u32 / u64 keeps the verdict bits in registers (v_cmp + v_addc_co per value, then an 8 x 8 butterfly between the
    column threads of a block: fl_consume.hpp compare_block_butterfly): every width -- the per-(W, row) code differs -- and all six
    predicates, constants inside, at and beyond the field's range; 13 blocks = one full and one partly filled wavefront, 1 block."""
    import operator
    ops = {"==": operator.eq, "!=": operator.ne, "<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}
    for ty in ("u32", "u64"):
        T = tbits(ty)
        for w in range(T + 1):
            for n in ((13, 1) if w in (0, 7, 17, T) else (13,)):
                pk = values(ty, n * packed_len(ty, w), 16000 + 64 * T + w)
                un = oracle.batch("unpack", ty, w, pk, n_blocks=n)
                dpk = to_dev(pk)
                maxv = (1 << w) - 1 if w else 0
                for k in sorted({0, maxv // 3, maxv, min(maxv + 1, (1 << T) - 1), int(un[7]), (1 << T) - 1}):
                    for name, f in ops.items():
                        got = fl.BitPacking.unpack_compare(w, dpk, name, k, n_blocks=n).cpu().numpy().view(np.uint8)
                        assert np.array_equal(got, np.packbits(f(un, TYPES[ty][0](k)), bitorder="little")), (ty, w, name, k, n)


def test_consumers_write_into_the_callers_output(fl, oracle):
    """unpack_compare / unpack_block_sums with output=: the caller's tensor is what gets written (and returned), guard elements on
    both sides stay, and a tensor of the wrong size / element width / tier is refused before any launch."""
    import torch
    n, w = 21, 11
    pk = values("u32", n * packed_len("u32", w), 4242)
    un = oracle.batch("unpack", "u32", w, pk, n_blocks=n)
    dpk = to_dev(pk)
    big = torch.full((n * 32 + 64,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    mask = big[32:32 + n * 32]
    got = fl.BitPacking.unpack_compare(w, dpk, "<=", 700, output=mask)
    assert got.data_ptr() == mask.data_ptr()
    assert np.array_equal(mask.cpu().numpy().view(np.uint8), np.packbits(un <= 700, bitorder="little"))
    assert bool((big[:32] == 0x5A5A5A5A).all()) and bool((big[32 + n * 32:] == 0x5A5A5A5A).all())
    sums = torch.full((n + 2,), -1, dtype=torch.int64, device="cuda")
    fl.BitPacking.unpack_block_sums(w, dpk, output=sums[1:1 + n])
    assert np.array_equal(sums[1:1 + n].cpu().numpy().view(np.uint64), un.reshape(n, 1024).astype(np.uint64).sum(axis=1))
    assert int(sums[0]) == -1 and int(sums[-1]) == -1
    for bad in (torch.empty(n * 32 - 1, dtype=torch.int32, device="cuda"), torch.empty(n * 32, dtype=torch.int64, device="cuda"),
                torch.empty(n * 32, dtype=torch.int32), torch.empty(2 * n * 32, dtype=torch.int32, device="cuda")[::2]):
        with pytest.raises((ValueError, TypeError)):
            fl.BitPacking.unpack_compare(w, dpk, "<=", 700, output=bad)
    with pytest.raises(ValueError):
        fl.BitPacking.unpack_block_sums(w, dpk, output=torch.empty(n + 1, dtype=torch.int64, device="cuda"))
    dun = to_dev(un)
    mm = torch.zeros(2 * n + 2, dtype=torch.uint32, device="cuda")
    mins, maxs = fl.BitPacking.block_min_max(dun, output=(mm[1:1 + n], mm[1 + n:1 + 2 * n]))
    assert mins.data_ptr() == mm[1:].data_ptr() and int(mm[0]) == 0 and int(mm[-1]) == 0
    assert np.array_equal(to_np(mins, "u32"), un.reshape(n, 1024).min(axis=1)) and np.array_equal(to_np(maxs, "u32"), un.reshape(n, 1024).max(axis=1))
    with pytest.raises(ValueError):
        fl.BitPacking.block_min_max(dun, output=(mm[:n], mm[n:2 * n + 1]))


def test_consumer_pair_places_the_output_by_probe(fl, oracle):
    """placement.consumer_pair: one allocation; the input inside a run of 8-GiB granules of one class of memory, the output in a
    granule of another class, classes found by the probe kernel (a small unpack_compare); the buffers it returns are usable as
    they are."""
    import torch
    from fastlanes_amd import placement as pl
    n, w = 1000, 9
    in_bytes, out_bytes = n * 128 * w, n * 128
    slab, src8, dst8, info = pl.consumer_pair(in_bytes, out_bytes, torch.device("cuda:0"), slab_bytes=64 << 30)
    assert slab.numel() == 64 << 30 and src8.numel() == in_bytes and dst8.numel() == out_bytes
    gi, go, classes = info["input_granule"], info["output_granule"], info["classes"]
    assert len(classes) == 8 and set(classes) <= set("ABC.") and classes[0] == "A"
    assert src8.data_ptr() == slab.data_ptr() + gi * pl.GRANULE_BYTES and dst8.data_ptr() == slab.data_ptr() + go * pl.GRANULE_BYTES and gi != go
    if "B" in classes:                       # two classes inside 64 GiB: the usual case, not a law
        assert info["input_one_class"] and classes[go] != classes[gi] and classes[go] != "."
    pk = values("u32", n * packed_len("u32", w), 321)
    src8.copy_(torch.from_numpy(pk.view(np.uint8)).to("cuda:0"))
    got = fl.BitPacking.unpack_compare(w, src8.view(torch.uint32), ">", 100, output=dst8.view(torch.int32))
    un = oracle.batch("unpack", "u32", w, pk, n_blocks=n)
    assert np.array_equal(got.cpu().numpy().view(np.uint8), np.packbits(un > 100, bitorder="little"))
    # the C ABI's probe (fl_internal_probe_memory_classes) draws the same map of the same allocation
    import ctypes
    n_granules = slab.numel() // pl.GRANULE_BYTES
    out = (ctypes.c_int * n_granules)()
    assert fl.load().fl_internal_probe_memory_classes(slab.data_ptr(), slab.numel(), out, None) == 0
    drawn = "".join("." if c < 0 else "ABC"[c] for c in out)
    # (an 8-GiB granule can straddle two classes -- they come in runs of GiB -- and then sits on the probe's threshold: one granule
    # may be judged differently by two runs of the probe, round 6)
    assert sum(a != b for a, b in zip(drawn, classes)) <= 1, (drawn, classes)
    # an output too large for one granule: the zone layout of column_pair
    slab2, s2, d2, info2 = pl.consumer_pair(1 << 20, 9 << 30, torch.device("cuda:0"))
    assert info2["output_granule"] is None and d2.numel() == 9 << 30
    del slab, slab2, s2, d2


@pytest.mark.parametrize("ty", TYS)
def test_unpack_compare_vs_oracle(fl, oracle, ty):
    """unpack_compare (extension): mask = numpy compare of the oracle's unpack
    output, packed LSB-first in index order; all six predicates incl. the k = 0 / k = MAX edges."""
    import operator
    T = tbits(ty)
    n = 41
    ops = {"==": operator.eq, "!=": operator.ne, "<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}
    # every width of the SWAR types (their kernels compare most fields in place, per-(W, row) code), a spread of the others
    widths = range(T + 1) if T <= 16 else sorted({0, 1, 3, 7, T // 2, T // 2 + 5, T - 1, T})
    for w in widths:
        pk = values(ty, n * packed_len(ty, w), 15000 + 64 * T + w)
        un = oracle.batch("unpack", ty, w, pk, n_blocks=n)
        dpk = to_dev(pk)
        maxv = (1 << w) - 1 if w else 0
        # incl. constants beyond the field's range (k > 2^W - 1: every x <= k, no x == k) and with the element's top bit set
        consts = sorted({0, 1, maxv // 2, maxv, min(maxv + 1, (1 << T) - 1), 1 << (T - 1), int(un[5]), (1 << T) - 1})
        for k in consts:
            for name, f in ops.items():
                got = fl.BitPacking.unpack_compare(w, dpk, name, k, n_blocks=n).cpu().numpy().view(np.uint8)
                want = np.packbits(f(un, TYPES[ty][0](k)), bitorder="little")
                assert np.array_equal(got, want), (ty, w, name, k)


@pytest.mark.parametrize("ty", TYS)
def test_host_tier_all_methods(fl, oracle, ty):
    """Every fl_<ty>_*_host entry point (numpy in / numpy out, the trait methods' own slices),
    several blocks per call and the single-block shape of one trait call."""
    T = tbits(ty)
    for n in (1, 3):
        for w in sorted({0, 1, T // 2 + 1, T}):
            v = values(ty, n * 1024, 800 + T + w)
            pk = values(ty, n * packed_len(ty, w), 801 + T + w)
            bases = values(ty, n * lanes(ty), 802 + T + w)
            ref = int(values(ty, 1, 803 + T + w)[0])
            assert np.array_equal(fl.BitPacking.pack(w, v), oracle.batch("pack", ty, w, v))
            assert np.array_equal(fl.BitPacking.unpack(w, pk, n_blocks=n), oracle.batch("unpack", ty, w, pk, n_blocks=n))
            refs = np.full(n, ref, dtype=TYPES[ty][0])
            assert np.array_equal(fl.FoR.for_pack(w, v, ref), oracle.batch("for_pack", ty, w, v, aux=refs))
            assert np.array_equal(fl.FoR.unfor_pack(w, pk, ref, n_blocks=n),
                                  oracle.batch("unfor_pack", ty, w, pk, aux=refs, n_blocks=n))
            assert np.array_equal(fl.Delta.undelta_pack(w, pk, bases),
                                  oracle.batch("undelta_pack", ty, w, pk, aux=bases, n_blocks=n))
            if w:
                i = (n - 1) * 1024 + 777
                pl = packed_len(ty, w)
                assert fl.BitPacking.unpack_single(w, pk, i, n_blocks=n) == \
                    oracle.unpack_single(ty, w, pk[(n - 1) * pl:n * pl], 777)
        v = values(ty, n * 1024, 900 + T)
        bases = values(ty, n * lanes(ty), 901 + T)
        assert np.array_equal(fl.Delta.delta(v, bases), oracle.batch("delta", ty, None, v, aux=bases))
        assert np.array_equal(fl.Delta.undelta(v, bases), oracle.batch("undelta", ty, None, v, aux=bases))
        assert np.array_equal(fl.Transpose.transpose(v), oracle.batch("transpose", ty, None, v))
        assert np.array_equal(fl.Transpose.untranspose(v), oracle.batch("untranspose", ty, None, v))


def test_concurrent_host_threads(fl, oracle):
    """The C ABI is re-entrant: host threads on their own streams (ctypes drops the GIL during
    the call) produce exact results concurrently, for different (type, width) instances."""
    import threading
    import torch
    jobs = [("u32", 7), ("u64", 17), ("u16", 3), ("u8", 5), ("u32", 12), ("u64", 33)]
    n = 400
    data = {j: values(j[0], n * packed_len(*j), 70 + i) for i, j in enumerate(jobs)}
    want = {j: oracle.batch("unpack", j[0], j[1], data[j]) for j in jobs}
    dev_in = {j: to_dev(data[j]) for j in jobs}
    errors = []

    def worker(j):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(25):
                    out = fl.BitPacking.unpack(j[1], dev_in[j])
                    back = fl.BitPacking.pack(j[1], out)
                s.synchronize()
            if not np.array_equal(to_np(out, j[0]), want[j]) or not np.array_equal(to_np(back, j[0]), data[j]):
                errors.append(j)
        except Exception as e:  # pragma: no cover
            errors.append((j, repr(e)))

    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_host_tier_threads_large_calls_and_release(fl, oracle):
    """The host tier keeps a per-thread cached context (private stream, pinned + device buffers): concurrent
    host threads get exact results, calls on either side of the zero-copy limit agree with the oracle, growing
    and releasing the cache (fl_host_release) is safe, and a thread's context dies with the thread."""
    import threading
    lib = fl.load()
    jobs = [("u32", 7, 1), ("u64", 17, 1), ("u16", 3, 1), ("u8", 5, 9), ("u32", 12, 300), ("u64", 33, 70)]
    data = {j: values(j[0], j[2] * packed_len(j[0], j[1]), 170 + i) for i, j in enumerate(jobs)}
    want = {j: oracle.batch("unpack", j[0], j[1], data[j]) for j in jobs}
    errors = []

    def worker(j):
        try:
            for it in range(40):
                out = fl.BitPacking.unpack(j[1], data[j])
                back = fl.BitPacking.pack(j[1], out)
                if not np.array_equal(out, want[j]) or not np.array_equal(back, data[j]):
                    errors.append((j, it))
                    return
                if it == 20:
                    lib.fl_host_release()            # the next call rebuilds the context
            i = (j[2] - 1) * 1024 + 513
            pl = packed_len(j[0], j[1])
            if fl.BitPacking.unpack_single(j[1], data[j], i) != oracle.unpack_single(j[0], j[1], data[j][(j[2] - 1) * pl:], 513):
                errors.append((j, "unpack_single"))
        except Exception as e:  # pragma: no cover
            errors.append((j, repr(e)))

    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # sizes straddling the zero-copy limit (256 KiB of in + out), then shrinking again
    for n in (1, 55, 56, 57, 2000, 3):
        pk = values("u32", n * 224, 190 + n)
        assert np.array_equal(fl.BitPacking.unpack(7, pk), oracle.batch("unpack", "u32", 7, pk)), n
        bases = values("u32", n * 32, 191 + n)
        assert np.array_equal(fl.Delta.undelta_pack(7, pk, bases), oracle.batch("undelta_pack", "u32", 7, pk, aux=bases, n_blocks=n)), n
    lib.fl_host_release()
    lib.fl_host_release()                            # idempotent


@pytest.mark.parametrize("ty", TYS)
def test_sixteen_byte_aligned_columns(fl, oracle, ty):
    """The documented precondition is 16-byte alignment (128 is only recommended): columns that
    start 16 / 48 bytes into an allocation decode and encode exactly."""
    import torch
    T = tbits(ty)
    esz = T // 8
    tdt = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}[ty]
    n = 45
    w = max(1, T // 3)
    pk = values(ty, n * packed_len(ty, w), 1234 + T)
    bases = values(ty, n * lanes(ty), 1235 + T)
    want_un = oracle.batch("unpack", ty, w, pk)
    want_ud = oracle.batch("undelta_pack", ty, w, pk, aux=bases)
    for off_in, off_out in ((16, 48), (48, 16)):
        raw_in = torch.zeros(pk.nbytes + 256, dtype=torch.uint8, device="cuda:0")
        raw_in[off_in:off_in + pk.nbytes] = torch.from_numpy(pk.view(np.uint8)).cuda()
        d_in = raw_in[off_in:off_in + pk.nbytes].view(tdt)
        raw_out = torch.zeros(n * 1024 * esz + 256, dtype=torch.uint8, device="cuda:0")
        d_out = raw_out[off_out:off_out + n * 1024 * esz].view(tdt)
        assert d_in.data_ptr() % 128 != 0 and d_out.data_ptr() % 128 != 0
        fl.BitPacking.unpack(w, d_in, output=d_out)
        assert np.array_equal(to_np(d_out, ty), want_un)
        fl.Delta.undelta_pack(w, d_in, to_dev(bases), output=d_out)
        assert np.array_equal(to_np(d_out, ty), want_ud)
        assert not raw_out[:off_out].any() and not raw_out[off_out + n * 1024 * esz:].any()
        # and back: pack from the offset unpacked column into an offset packed column
        fl.BitPacking.unpack(w, d_in, output=d_out)
        raw_pk2 = torch.zeros(pk.nbytes + 256, dtype=torch.uint8, device="cuda:0")
        d_pk2 = raw_pk2[off_out:off_out + pk.nbytes].view(tdt)
        fl.BitPacking.pack(w, d_out, output=d_pk2)
        assert np.array_equal(to_np(d_pk2, ty), oracle.batch("pack", ty, w, want_un))


def _splitmix_on_device(n_words, seed):
    """The stream fl_oracle_parallel_fill writes (splitmix64 of seed + (i+1)*GOLDEN), regenerated on
    the GPU with wrapping int64 arithmetic -- host and device see identical bits without a PCIe copy."""
    import torch
    def c(x):  # two's-complement view of a uint64 constant
        return x - (1 << 64) if x >= (1 << 63) else x
    G, C1, C2 = c(0x9E3779B97F4A7C15), c(0xBF58476D1CE4E5B9), c(0x94D049BB133111EB)
    z = torch.arange(1, n_words + 1, dtype=torch.int64, device="cuda:0") * G + c(seed & (2**64 - 1))
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * C1       # logical >> via mask
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * C2
    return z ^ ((z >> 31) & ((1 << 33) - 1))


@pytest.mark.parametrize("ty,w,op", [("u32", 7, "unpack"), ("u64", 17, "unpack"), ("u32", 12, "undelta_pack")])
def test_full_column_content_hash_vs_cpu_oracle(fl, oracle, ty, w, op):
    """SURVEY.md 8(d) 'correctness at scale' (2) for BASELINE configs 2, 3 and 4: the whole 10 M-block
    column, chunk by chunk -- the GPU decodes a device-generated stream, the multithreaded CPU oracle
    decodes the same stream regenerated on the host, and two 64-bit content hashes per block (sum and
    position-weighted sum, wrapping) of ALL 10.24 G values must agree."""
    import torch
    n_total, chunk = 10_000_000, 1_000_000
    T = tbits(ty)
    tdt = {"u32": torch.uint32, "u64": torch.uint64}[ty]
    wpb = 128 * w // 8                      # 64-bit words per packed block
    threads = min(64, os.cpu_count() or 1)
    w_idx = torch.arange(1, 1025, dtype=torch.int64, device="cuda:0")
    for k in range(n_total // chunk):
        seed = 0xC0FFEE + 17 * T + k
        host_pk = np.empty(chunk * packed_len(ty, w), dtype=TYPES[ty][0])
        oracle.parallel_fill(host_pk, 128 * w, chunk, seed, threads)
        dev_pk = _splitmix_on_device(chunk * wpb, seed).view(tdt)
        if k == 0:   # the two generators really are the same stream
            assert np.array_equal(to_np(dev_pk[:4096], ty), host_pk[:4096])
        if op == "undelta_pack":
            host_bases = np.empty(chunk * lanes(ty), dtype=TYPES[ty][0])
            oracle.parallel_fill(host_bases, 128, chunk, seed + 1000, threads)
            dev_bases = _splitmix_on_device(chunk * 16, seed + 1000).view(tdt)
            host_out = oracle.fast(op, ty, w, host_pk, aux=host_bases, n_blocks=chunk, nthreads=threads)
            out = fl.Delta.undelta_pack(w, dev_pk, dev_bases)
        else:
            host_out = oracle.fast(op, ty, w, host_pk, n_blocks=chunk, nthreads=threads)
            out = fl.BitPacking.unpack(w, dev_pk)
        s_cpu, w_cpu = oracle.block_hashes(ty, host_out, threads)
        if ty == "u32":
            vals = out.view(torch.int32).view(chunk, 1024).to(torch.int64) & 0xFFFFFFFF
        else:
            vals = out.view(torch.int64).view(chunk, 1024)
        s_gpu = vals.sum(dim=1)
        w_gpu = (vals * w_idx).sum(dim=1)
        assert np.array_equal(s_gpu.cpu().numpy().view(np.uint64), s_cpu), k
        assert np.array_equal(w_gpu.cpu().numpy().view(np.uint64), w_cpu), k
        del vals, out, dev_pk


def test_config5_full_column_content_hash_vs_cpu_oracle(fl, oracle):
    """SURVEY.md 8(d) 'correctness at scale' (2) for BASELINE config 5: ALL 10 B integers of the mixed-width column
    (u32, width[b] = 1 + b mod 32, 9 765 625 blocks), slice by slice of the 8-GPU sharding.  The GPU decodes a
    device-generated stream with device-resident widths / offsets; the multithreaded CPU oracle runs the reference's
    caller loop over the same stream regenerated on the host; two 64-bit content hashes per block must agree."""
    import torch
    from fastlanes_amd.sharding import block_range
    n_total = 9_765_625
    threads = min(64, os.cpu_count() or 1)
    w_idx = torch.arange(1, 1025, dtype=torch.int64, device="cuda:0")
    done = 0
    for r in range(8):
        first, n = block_range(n_total, 8, r)
        widths = (1 + (np.arange(n, dtype=np.int64) + first) % 32).astype(np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(widths.astype(np.uint64) * np.uint64(128), out=off[1:])
        nwords = int(off[-1]) // 8
        seed = 0xC5 + 131 * r
        host_pk = np.empty(nwords * 2, dtype=np.uint32)
        oracle.parallel_fill(host_pk, 8, nwords, seed, threads)
        dev_pk = _splitmix_on_device(nwords, seed).view(torch.uint32)
        if r == 0:
            assert np.array_equal(to_np(dev_pk[:4096], "u32"), host_pk[:4096])
        dw = torch.from_numpy(widths).cuda()
        doff, dtotal = fl.widths_to_offsets("u32", dw)
        assert int(dtotal.item()) == int(off[-1])
        out = fl.unpack_widths(dw, doff, dev_pk)
        host_out = oracle.fast_unpack_mixed_u32(widths, off[:-1], host_pk, nthreads=threads)
        s_cpu, w_cpu = oracle.block_hashes("u32", host_out, threads)
        vals = out.view(torch.int32).view(n, 1024).to(torch.int64) & 0xFFFFFFFF
        assert np.array_equal(vals.sum(dim=1).cpu().numpy().view(np.uint64), s_cpu), r
        assert np.array_equal((vals * w_idx).sum(dim=1).cpu().numpy().view(np.uint64), w_cpu), r
        done += n
        del vals, out, dev_pk, host_out
    assert done == n_total


def test_for_reference_stride_through_the_c_abi(fl, oracle):
    """fl_<ty>_for_pack / unfor_pack read references[b * reference_stride]: 0 broadcasts one scalar,
    1 is one per block, larger strides pick every k-th element (e.g. a struct-of-stats array)."""
    import ctypes
    import torch
    lib = fl.load()
    n, w = 70, 9
    v = values("u32", n * 1024, 31)
    refs3 = values("u32", n * 3, 32)
    dv, dr = to_dev(v), to_dev(refs3)
    out = torch.empty(n * packed_len("u32", w), dtype=torch.uint32, device="cuda:0")
    for stride, eff in ((0, np.full(n, refs3[0])), (1, refs3[:n]), (3, refs3[::3])):
        rc = lib.fl_u32_for_pack(w, dv.data_ptr(), dr.data_ptr(), stride, out.data_ptr(), n, None)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(to_np(out, "u32"), oracle.batch("for_pack", "u32", w, v, aux=eff)), stride
        back = torch.empty(n * 1024, dtype=torch.uint32, device="cuda:0")
        assert lib.fl_u32_unfor_pack(w, out.data_ptr(), dr.data_ptr(), stride, back.data_ptr(), n, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(to_np(back, "u32"), oracle.batch("unfor_pack", "u32", w, to_np(out, "u32"), aux=eff)), stride


@pytest.mark.parametrize("backend", ["gloo", "auto"])
def test_bench_two_ranks_on_this_gpu(fl, backend):
    """The N > 1 path of bench.py with the real kernels: `bench.py --gpus 2` spawns two ranks (both on this GPU), each
    decodes its own weak-scaled column and its half of the strong-scaled 10 B-integer mixed-width column and checks the
    first / last / sampled blocks of ITS slice against the oracle; rank 0 prints one line with both ranks' timings and
    verdicts.  backend=auto tries RCCL first: two ranks on one device is something RCCL refuses, so this is the
    RCCL-failure path on real hardware -- the run must finish on gloo and say why."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", backend, "--single-device",
                        "--nccl-probe-timeout", "60", "--steps", "3", "--warmup", "1", "--blocks", "500000"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert [p["rank"] for p in d["per_rank"]] == [0, 1] and all(p["blocks"] == 500000 and p["GBps"] > 0 for p in d["per_rank"])
    assert [p["correct"] for p in d["per_rank"]] == [True, True] and "bit-exact" in d["correctness"]
    c5 = d["config5_strong"]
    assert c5["scaling"] == "strong" and c5["n_gpus"] == 2 and c5["value"] > 0
    assert [p["blocks"] for p in c5["per_rank"]] == [4882813, 4882812] and c5["per_rank"][1]["first_block"] == 4882813
    assert [p["correct"] for p in c5["per_rank"]] == [True, True]
    if backend == "gloo":
        assert d["control_backend"] == "gloo" and "control_fallback_reason" not in d
    else:
        assert d["control_backend"] in ("gloo", "nccl")
        if d["control_backend"] == "gloo":
            assert d["control_fallback_reason"]


# ---------------------------------------------------------------------------
# Under load.  The per-(T, W) tests above run tens of blocks: a handful of wavefronts on an idle chip.  A kernel can be right
# there and wrong with every CU busy (round 3: a register-only form of Delta's lane-group scan passed all of them and produced
# 3 % wrong blocks, differently on every run, at 735 000 blocks).  So: every kernel family at a column size that fills the chip
# many times over -- the result must be the same on two runs, and right (against the oracle) on the first / last blocks and a
# seeded sample of the rest.
# ---------------------------------------------------------------------------
UNDER_LOAD_WIDTHS = {"u8": (3, 8), "u16": (9,), "u32": (7, 12, 20), "u64": (17, 20)}


def _sampled_block_indices(n, k, seed):
    rng = np.random.default_rng(seed)
    return np.unique(np.concatenate([np.arange(4), np.arange(n - 4, n), rng.integers(0, n, size=k)]))


def _gather(t, idx, per_block):
    """blocks idx of a flat device tensor -> one contiguous numpy array"""
    import torch
    i = torch.from_numpy(idx).to(t.device)
    return t.view(torch.uint8).view(-1, per_block * t.element_size())[i].contiguous().cpu().numpy().reshape(-1)      # (torch indexes uint8, not uint16/32/64)


@pytest.mark.parametrize("ty", TYS)
def test_every_kernel_family_under_load(fl, oracle, ty):
    import torch
    T = tbits(ty)
    L = lanes(ty)
    dt = TYPES[ty][0]
    tdt = getattr(torch, str(np.dtype(dt)))
    n = 600_000 if T <= 16 else 300_000
    lib = fl.load()
    idx = _sampled_block_indices(n, 120, 99 + T)

    def filled(n_elems, seed):
        t = torch.empty(n_elems, dtype=tdt, device="cuda:0")
        assert lib.fl_fill_random(t.data_ptr(), (n_elems * (T // 8)) & ~7, seed, None) == 0
        return t

    def twice(f):
        a = f().clone()
        b = f()
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), "two runs of the same call differ"
        return a

    un = filled(n * 1024, 5)
    bases = filled(n * L, 6)
    refs = filled(n, 7)
    un_s = _gather(un, idx, 1024).view(dt)
    bases_s = _gather(bases, idx, L).view(dt)
    refs_s = _gather(refs, idx, 1).view(dt)
    k = len(idx)
    # the width-free kernels
    for name, call in (("delta", lambda: fl.Delta.delta(un, bases)), ("undelta", lambda: fl.Delta.undelta(un, bases)),
                       ("transpose", lambda: fl.Transpose.transpose(un)), ("untranspose", lambda: fl.Transpose.untranspose(un))):
        got = _gather(twice(call), idx, 1024).view(dt)
        want = oracle.batch(name, ty, None, un_s, aux=bases_s) if "delta" in name else oracle.batch(name, ty, None, un_s)
        assert np.array_equal(got, want), (ty, name)
    mins, maxs = fl.BitPacking.block_min_max(un)
    assert np.array_equal(to_np(mins, ty)[idx], un_s.reshape(k, 1024).min(axis=1)) and np.array_equal(to_np(maxs, ty)[idx], un_s.reshape(k, 1024).max(axis=1))
    for w in UNDER_LOAD_WIDTHS[ty]:
        pl = packed_len(ty, w)
        pk = filled(n * pl, 11 + w)
        pk_s = _gather(pk, idx, pl).view(dt)
        unpacked_s = oracle.batch("unpack", ty, w, pk_s, n_blocks=k)
        checks = (
            ("unpack", lambda: fl.BitPacking.unpack(w, pk), 1024, lambda: unpacked_s),
            ("pack", lambda: fl.BitPacking.pack(w, un), pl, lambda: oracle.batch("pack", ty, w, un_s)),
            ("unfor_pack", lambda: fl.FoR.unfor_pack(w, pk, refs), 1024, lambda: oracle.batch("unfor_pack", ty, w, pk_s, aux=refs_s, n_blocks=k)),
            ("for_pack", lambda: fl.FoR.for_pack(w, un, refs), pl, lambda: oracle.batch("for_pack", ty, w, un_s, aux=refs_s)),
            ("undelta_pack", lambda: fl.Delta.undelta_pack(w, pk, bases), 1024, lambda: oracle.batch("undelta_pack", ty, w, pk_s, aux=bases_s, n_blocks=k)),
            ("undelta_pack_untranspose", lambda: fl.Delta.undelta_pack_untranspose(w, pk, bases), 1024,
             lambda: oracle.batch("untranspose", ty, None, oracle.batch("undelta_pack", ty, w, pk_s, aux=bases_s, n_blocks=k))),
            ("transpose_delta_pack", lambda: fl.Delta.transpose_delta_pack(w, un, bases), pl,
             lambda: oracle.batch("pack", ty, w, oracle.batch("delta", ty, None, oracle.batch("transpose", ty, None, un_s), aux=bases_s))),
        )
        for name, call, per_block, want in checks:
            got = _gather(twice(call), idx, per_block).view(dt)
            assert np.array_equal(got, want()), (ty, w, name)
        kc = (1 << w) // 3
        mask = twice(lambda: fl.BitPacking.unpack_compare(w, pk, "<=", kc))
        assert np.array_equal(_gather(mask, idx, 32), np.packbits(unpacked_s <= dt(kc), bitorder="little")), (ty, w, "unpack_compare")
        sums = twice(lambda: fl.BitPacking.unpack_block_sums(w, pk))
        assert np.array_equal(sums.cpu().numpy().view(np.uint64)[idx], unpacked_s.reshape(k, 1024).astype(np.uint64).sum(axis=1)), (ty, w, "sums")
        del pk


def test_column_pair_alloc_and_the_bare_stream(fl, oracle):
    """fl_column_pair_alloc (the C ABI's optional allocation helper): every layout hands out buffers the codec works in, the
    zoned layout is what the header says, the probe reports what it measured; fl_internal_bare_stream (bench.py's yardstick)
    touches exactly its units."""
    import ctypes
    import torch
    from fastlanes_amd import placement as pl
    lib = fl.load()
    n, W = 20011, 7
    pk = values("u32", n * packed_len("u32", W), 9100)
    want = oracle.batch("unpack", "u32", W, pk)
    for layout in ("separate", "zoned", "auto", "interleaved"):      # (a pair this small is plain allocations under "interleaved")
        pair = pl.ColumnPair(n * 128 * W, n * 4096, "cuda:0", aux_bytes=n * 128, layout=layout)
        assert pair.layout in ("separate", "zoned", "interleaved") and (layout == "auto" or pair.layout == layout)
        for t, nb in ((pair.input, n * 128 * W), (pair.aux, n * 128), (pair.output, n * 4096)):
            assert t.numel() == nb and t.data_ptr() % 256 == 0 and t.device.index == 0
        if pair.layout == "zoned":       # input at the start, aux behind it, the output centred on the 64-GiB multiple
            assert pair.aux.data_ptr() - pair.input.data_ptr() == (n * 128 * W + 255) // 256 * 256
            half = ((n * 4096 + 255) // 256 * 256) // 2
            assert pair.output.data_ptr() - pair.input.data_ptr() == ((64 << 30) - half) // 256 * 256
        if layout == "auto":
            assert pair.probe_GBps and set(pair.probe_GBps) <= {"separate", "zoned"} and all(v > 50 for v in pair.probe_GBps.values())
            assert pair.layout == max(pair.probe_GBps, key=pair.probe_GBps.get)
        else:
            assert pair.probe_GBps is None
        pair.input.view(torch.uint32).copy_(to_dev(pk))
        got = fl.BitPacking.unpack(W, pair.input.view(torch.uint32), output=pair.output.view(torch.uint32))
        assert np.array_equal(to_np(got, "u32"), want), layout
        pair.free()
        pair.free()                      # idempotent
    # the bare stream: n units of 896 B (+ 128 B aux) read, 4096 B written; nothing outside them is touched, ragged unit count
    src = to_dev(values("u8", n * 896, 9101))
    aux = to_dev(values("u8", n * 128, 9102))
    out = torch.full((n * 4096 + 8192,), 0x5A, dtype=torch.uint8, device="cuda:0")
    for nt, waves, window in ((0, 5, 31), (1, 8, 16), (1, 3, 12)):
        out.fill_(0x5A)
        assert lib.fl_internal_bare_stream(src.data_ptr(), 896, aux.data_ptr(), 128, out.data_ptr(), 4096, n, nt, waves, window, None) == 0
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        assert (o[n * 4096:] == 0x5A).all() and not (o[:n * 4096].reshape(n, 4096) == 0x5A).all(axis=1).any()
        if nt == 0:
            first = o[:n * 4096].copy()
        else:
            assert np.array_equal(o[:n * 4096], first)          # the tile map and the cache policy never change a byte
    # a pack-shaped stream (8 KiB read : 2176 B written) and a read-only one
    big = to_dev(values("u8", 1000 * 8192, 9103))
    small = torch.zeros(1000 * 2176 + 4096, dtype=torch.uint8, device="cuda:0")
    assert lib.fl_internal_bare_stream(big.data_ptr(), 8192, None, 0, small.data_ptr(), 2176, 1000, 1, 8, 16, None) == 0
    torch.cuda.synchronize()
    assert not small[1000 * 2176:].any().item() and small[:1000 * 2176].any().item()
    small.zero_()
    assert lib.fl_internal_bare_stream(big.data_ptr(), 8192, None, 0, small.data_ptr(), 0, 1000, 1, 8, 31, None) == 0
    torch.cuda.synchronize()
    assert not small.any().item()


@pytest.mark.parametrize("n_blocks", [2_000_000])
def test_interleaved_column_pair_is_constructed_from_measured_chunks(fl, oracle, n_blocks):
    """FL_LAYOUT_INTERLEAVED (round 6): the pair is built from 1-GiB physical chunks whose class of memory
    was measured; input + aux inside one class, the output's chunks arranged for the eight XCDs' write positions.  The codec decodes in it exactly as in
    plain allocations (oracle on the blocks around every chunk boundary, the whole output against a plain-allocation decode), a second
    pair never gets the first one's addresses (this ROCm keeps stale translations for re-used ranges: tools/exp_vmm remap), and
    "auto" reports the constructed layout's figure next to the others."""
    import torch
    from fastlanes_amd import placement as pl
    n, W = n_blocks, 7
    ib, ob, ab = n * 128 * W, n * 4096, n * 128
    chunk = 1 << 30
    g = torch.Generator(device="cuda:0").manual_seed(n)
    pk = torch.randint(0, 1 << 31, (ib // 4,), dtype=torch.int32, device="cuda:0", generator=g).view(torch.uint32)
    plain = fl.BitPacking.unpack(W, pk)
    pairs = []
    for rep in range(2):
        pair = pl.ColumnPair(ib, ob, "cuda:0", aux_bytes=ab, layout="interleaved")
        assert pair.layout == "interleaved" and pair.probe_GBps is None
        n_out = -(-ob // chunk)
        n_in = -(-(((ib + 255) // 256 * 256) + ((ab + 255) // 256 * 256)) // chunk)
        assert len(pair.classes) == min(95, n_in + n_out), (pair.classes, n_in, n_out)
        assert pair.output.data_ptr() - pair.input.data_ptr() == n_in * chunk and pair.aux.data_ptr() - pair.input.data_ptr() == (ib + 255) // 256 * 256
        cin, cout = pair.classes[:n_in], pair.classes[n_in:]
        if set(pair.classes) >= {"A", "B", "C"}:           # three classes seen: the construction the header promises
            assert len(set(cin)) == 1 and len(set(cout)) >= 2, pair.classes
            assert sum(ch != cin[0] for ch in cout) * 2 >= len(cout), pair.classes      # mostly the OTHER classes
        pair.input.view(torch.uint32).copy_(pk)
        pair.output.fill_(0xEE)
        got = fl.BitPacking.unpack(W, pair.input.view(torch.uint32), output=pair.output.view(torch.uint32))
        assert torch.equal(got, plain)
        # the oracle on the blocks either side of every chunk boundary of the output, and of the input
        edges = sorted({0, n - 1} | {min(n - 1, max(0, (k * chunk) // 4096 + d)) for k in range(1, n_out + 1) for d in (-1, 0)}
                       | {min(n - 1, max(0, (k * chunk) // (128 * W) + d)) for k in range(1, n_in + 1) for d in (-1, 0, 1)})
        host_pk = to_np(pk, "u32").reshape(n, 32 * W)
        host_out = to_np(got, "u32").reshape(n, 1024)
        want = oracle.batch("unpack", "u32", W, np.ascontiguousarray(host_pk[edges]).reshape(-1)).reshape(len(edges), 1024)
        assert np.array_equal(host_out[edges], want)
        pairs.append((pair.input.data_ptr(), pair.output.data_ptr() + ob))
        pair.free()
    (a0, a1), (b0, b1) = pairs
    assert b0 >= a1 or b1 <= a0, "an interleaved pair re-used a freed pair's addresses"
    # the encode direction: a read-dominated pair's output rotates through ALL three classes, and calls whose buffers lie inside one live
    # constructed pair launch under the whole-column tile map (fl_kernels.hpp: constructed_pair_this_thread) -- the same bytes as in plain
    # tensors, for pack and for a 1 : 1 stream (transpose), and the plain-tensor calls in between are unaffected
    vals = fl.BitPacking.unpack(W, pk)
    want_pk, want_tr = fl.BitPacking.pack(W, vals), fl.Transpose.transpose(vals)
    enc = pl.ColumnPair(ob, ib, "cuda:0", layout="interleaved")
    assert enc.layout == "interleaved" and len(enc.classes) == n_out + -(-ib // chunk)
    if set(enc.classes) >= {"A", "B", "C"}:
        assert len(set(enc.classes[:n_out])) == 1, enc.classes
    enc.input.view(torch.uint32).copy_(vals)
    enc.output.fill_(0xEE)
    assert torch.equal(fl.BitPacking.pack(W, enc.input.view(torch.uint32), output=enc.output.view(torch.uint32)), want_pk)
    assert torch.equal(fl.BitPacking.pack(W, vals), want_pk)
    enc.free()
    tr = pl.ColumnPair(ob, ob, "cuda:0", layout="interleaved")
    tr.input.view(torch.uint32).copy_(vals)
    assert torch.equal(fl.Transpose.transpose(tr.input.view(torch.uint32), output=tr.output.view(torch.uint32)), want_tr)
    assert torch.equal(fl.Transpose.untranspose(tr.output.view(torch.uint32), output=tr.input.view(torch.uint32)), vals)
    tr.free()
    del vals, want_pk, want_tr
    auto = pl.ColumnPair(ib, ob, "cuda:0", layout="auto")
    assert "interleaved" in auto.probe_GBps and "separate" in auto.probe_GBps and all(v > 1000 for v in auto.probe_GBps.values())
    best = max(auto.probe_GBps.values())
    assert auto.probe_GBps[auto.layout] >= 0.98 * best / 1.02, (auto.layout, auto.probe_GBps)
    auto.free()


def test_zero_copy_host_calls_under_load_never_fall_back(fl, oracle):
    """The host tier's small calls are zero-copy and wait for a completion word in pinned memory instead of synchronising the
    stream (fl_capi.hip: HostCtx::wait_zero_copy): with the chip kept busy by a device-tier stream, 1500 single-block trait calls per
    type return exactly the oracle's bytes, and not one of them hit the 50 ms fallback (fl_internal_zero_copy_fallbacks)."""
    import threading
    import torch
    lib = fl.load()
    before = lib.fl_internal_zero_copy_fallbacks()
    big_in = to_dev(values("u32", 400000 * packed_len("u32", 7), 9200))
    big_out = torch.empty(400000 * 1024, dtype=torch.uint32, device="cuda:0")
    stop = threading.Event()

    def load():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop.is_set():
                fl.BitPacking.unpack(7, big_in, output=big_out)
                s.synchronize()
    t = threading.Thread(target=load)
    t.start()
    try:
        for ty, w in (("u32", 7), ("u64", 17), ("u16", 3), ("u8", 5)):
            pl = packed_len(ty, w)
            v = values(ty, 1500 * 1024, 9201 + w)
            for b in range(1500):
                blk = v[b * 1024:(b + 1) * 1024]
                pk = fl.BitPacking.pack(w, blk)
                assert pk.shape[0] == pl
                if b % 50 == 0:
                    assert np.array_equal(pk, oracle.pack(ty, w, blk)), (ty, b)
                back = fl.BitPacking.unpack(w, pk)
                m = np.array((1 << w) - 1, dtype=np.uint64).astype(TYPES[ty][0])
                assert np.array_equal(back, blk & m), (ty, b)
    finally:
        stop.set()
        t.join()
    assert lib.fl_internal_zero_copy_fallbacks() == before


@pytest.mark.gpu
def test_two_threads_construct_pairs_at_once(fl, oracle):
    """fl_column_pair_alloc(FL_LAYOUT_INTERLEAVED) from two host threads at the same time (the class probe times small kernels: constructions
    are serialised inside the library); both pairs are constructed, disjoint, and decode exactly like plain tensors."""
    import threading
    import torch
    from fastlanes_amd import placement as pl
    n, W = 2_000_000, 7
    ib, ob = n * 128 * W, n * 4096
    g = torch.Generator(device="cuda:0").manual_seed(77)
    pk = torch.randint(0, 1 << 31, (ib // 4,), dtype=torch.int32, device="cuda:0", generator=g).view(torch.uint32)
    want = fl.BitPacking.unpack(W, pk)
    pairs, errors = [None, None], []

    def build(k):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                pairs[k] = pl.ColumnPair(ib, ob, "cuda:0", layout="interleaved")
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=build, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errors, errors
    a, b = pairs
    assert a.layout == b.layout == "interleaved" and a.classes and b.classes
    lo_a, hi_a = a.input.data_ptr(), a.output.data_ptr() + ob
    lo_b, hi_b = b.input.data_ptr(), b.output.data_ptr() + ob
    assert hi_a <= lo_b or hi_b <= lo_a
    for p in pairs:
        p.input.view(torch.uint32).copy_(pk)
        assert torch.equal(fl.BitPacking.unpack(W, p.input.view(torch.uint32), output=p.output.view(torch.uint32)), want)
        p.free()
