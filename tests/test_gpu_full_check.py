"""Full-output verification under load (pytest -m gpu).

tests/test_gpu_parity.py compares the kernels with the CPU oracle on tens of blocks, and -- at sizes that fill the chip -- on a
SAMPLE of blocks (test_every_kernel_family_under_load).  Round 3 met a kernel that was right on an idle chip and wrong on a few
per cent of the blocks, differently on every run, with all CUs busy (profiles/abscan_r03.txt): a deterministic error confined
to blocks nobody samples would still pass.  So here EVERY element of the output of EVERY kernel family is verified at >= 500 000
blocks per call, for all four types, under the three kernel policies, while a second stream keeps the chip busy -- by a
deliberately naive device-side checker (tests/checker/naive_check.hip: one thread per value, the reference's closed forms,
no code shared with fastlanes_amd/csrc) that counts differing elements.  The checker itself is first validated against the
CPU oracle at small sizes: the oracle's output must count 0 differences, a single corrupted element exactly 1."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from datagen import values
from oracle_lib import TYPES, lanes, packed_len, tbits

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYS = ["u8", "u16", "u32", "u64"]
OPS = {"unpack": 0, "pack": 1, "delta": 2, "undelta": 3, "undelta_pack": 4, "transpose": 5, "untranspose": 6,
       "undelta_pack_untranspose": 7, "transpose_delta_pack": 8, "block_sums": 9, "compare": 10, "min_max": 11}
CMP = {"==": 0, "!=": 1, "<": 2, "<=": 3, ">": 4, ">=": 5}


def build_checker():
    src = os.path.join(ROOT, "tests", "checker", "naive_check.hip")
    so = os.path.join(ROOT, "tests", "checker", "libfl_naive_check.so")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
    return so


class Checker:
    def __init__(self):
        import torch
        self.torch = torch
        self.lib = ctypes.CDLL(build_checker())
        P, Q = ctypes.c_void_p, ctypes.c_uint64
        self.lib.naive_check.restype = ctypes.c_int
        self.lib.naive_check.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_uint, P, P, Q, P, P, Q, P, ctypes.c_int, Q, P, P, P]
        self.counter = torch.zeros(1, dtype=torch.int64, device="cuda:0")

    def mismatches(self, ty, op, width, a, got, n_blocks, aux=None, aux_stride=0, got2=None, cmp_op="==", cmp_k=0, widths=None, offsets=None):
        """number of elements of `got` that differ from what the closed forms say"""
        torch = self.torch
        self.counter.zero_()
        ptr = lambda t: None if t is None else t.data_ptr()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = self.lib.naive_check(tbits(ty), OPS[op], width, ptr(a), ptr(aux), aux_stride, ptr(got), ptr(got2), n_blocks,
                                  self.counter.data_ptr(), CMP[cmp_op], int(cmp_k), ptr(widths), ptr(offsets), st)
        assert rc == 0, f"naive_check failed to launch: {rc}"
        return int(self.counter.item())


@pytest.fixture(scope="module")
def checker():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return Checker()


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
    return torch.from_numpy(a.view(np.uint8)).to("cuda:0").view(getattr(torch, str(a.dtype)))


def corrupt(t, at):
    """flip the lowest bit of element `at` of a device tensor; returns the corrupted copy"""
    import torch
    c = t.clone()
    b = c.view(torch.uint8)
    b[at * t.element_size()] ^= 1
    return c


@pytest.mark.parametrize("ty", TYS)
def test_the_checker_itself_against_the_oracle(checker, oracle, ty):
    """The checker is written from the reference's closed forms (unpack_single, index, transpose; per-lane serial Delta), not
    from the kernels.  Before it judges them: for every op and every width the ORACLE's output must count 0 differences and the
    same output with ONE element corrupted exactly 1 -- right, and sensitive."""
    T, L = tbits(ty), lanes(ty)
    n = 9
    v = values(ty, n * 1024, 400 + T)
    bases = values(ty, n * L, 401 + T)
    refs = values(ty, n, 402 + T)
    dv, db, dr = to_dev(v), to_dev(bases), to_dev(refs)
    at = 5 * 1024 + 321
    for op in ("delta", "undelta", "transpose", "untranspose"):
        want = to_dev(oracle.batch(op, ty, None, v, aux=bases) if "delta" in op else oracle.batch(op, ty, None, v))
        assert checker.mismatches(ty, op, 0, dv, want, n, aux=db) == 0, (ty, op)
        assert checker.mismatches(ty, op, 0, dv, corrupt(want, at), n, aux=db) == 1, (ty, op)
    mn, mx = to_dev(v.reshape(n, 1024).min(axis=1)), to_dev(v.reshape(n, 1024).max(axis=1))
    assert checker.mismatches(ty, "min_max", 0, dv, mn, n, got2=mx) == 0
    assert checker.mismatches(ty, "min_max", 0, dv, mn, n, got2=corrupt(mx, 3)) == 1
    for w in range(T + 1):
        pl = packed_len(ty, w)
        pk = values(ty, n * pl, 500 + 64 * T + w)
        dpk = to_dev(pk) if pl else to_dev(np.zeros(1, dtype=v.dtype))
        un = oracle.batch("unpack", ty, w, pk, n_blocks=n)
        for op, src, want, kw in (
                ("unpack", dpk, un, {}),
                ("unpack", dpk, oracle.batch("unfor_pack", ty, w, pk, aux=refs, n_blocks=n), dict(aux=dr, aux_stride=1)),
                ("undelta_pack", dpk, oracle.batch("undelta_pack", ty, w, pk, aux=bases, n_blocks=n), dict(aux=db)),
                ("undelta_pack_untranspose", dpk, oracle.batch("untranspose", ty, None, oracle.batch("undelta_pack", ty, w, pk, aux=bases, n_blocks=n)),
                 dict(aux=db))):
            dw = to_dev(want)
            assert checker.mismatches(ty, op, w, src, dw, n, **kw) == 0, (ty, w, op)
            assert checker.mismatches(ty, op, w, src, corrupt(dw, at), n, **kw) == 1, (ty, w, op)
        if w:
            for op, want, kw in (
                    ("pack", oracle.batch("pack", ty, w, v), {}),
                    ("pack", oracle.batch("for_pack", ty, w, v, aux=refs), dict(aux=dr, aux_stride=1)),
                    ("transpose_delta_pack", oracle.batch("pack", ty, w, oracle.batch("delta", ty, None, oracle.batch("transpose", ty, None, v), aux=bases)),
                     dict(aux=db))):
                dw = to_dev(want)
                assert checker.mismatches(ty, op, w, dv, dw, n, **kw) == 0, (ty, w, op)
                assert checker.mismatches(ty, op, w, dv, corrupt(dw, 4 * pl + 7), n, **kw) == 1, (ty, w, op)     # one bit = one field
        sums = to_dev(un.reshape(n, 1024).astype(np.uint64).sum(axis=1))
        assert checker.mismatches(ty, "block_sums", w, dpk, sums, n) == 0, (ty, w)
        assert checker.mismatches(ty, "block_sums", w, dpk, corrupt(sums, 2), n) == 1, (ty, w)
        k = int(un[77]) if w else 0
        for name, f in (("<=", np.less_equal), ("==", np.equal), (">", np.greater)):
            mask = to_dev(np.packbits(f(un, TYPES[ty][0](k)), bitorder="little").view(np.uint32))
            assert checker.mismatches(ty, "compare", w, dpk, mask, n, cmp_op=name, cmp_k=k) == 0, (ty, w, name)
            assert checker.mismatches(ty, "compare", w, dpk, corrupt(mask, 40), n, cmp_op=name, cmp_k=k) == 1, (ty, w, name)
    # mixed widths: per-block widths / byte offsets
    widths = (np.arange(n) * 5 % (T + 1)).astype(np.uint8)
    off = np.concatenate([[0], np.cumsum(widths.astype(np.uint64) * 128)]).astype(np.uint64)
    col = values(ty, max(1, int(off[-1]) // (T // 8)), 600 + T)
    want = np.concatenate([oracle.unpack(ty, int(w), col[int(off[b]) // (T // 8):int(off[b + 1]) // (T // 8)]) for b, w in enumerate(widths)])
    dcol, dw_, doff = to_dev(col), to_dev(widths), to_dev(off[:-1])
    assert checker.mismatches(ty, "unpack", 0, dcol, to_dev(want), n, widths=dw_, offsets=doff) == 0
    assert checker.mismatches(ty, "unpack", 0, dcol, corrupt(to_dev(want), at), n, widths=dw_, offsets=doff) == 1
    want = np.concatenate([oracle.undelta_pack(ty, int(w), col[int(off[b]) // (T // 8):int(off[b + 1]) // (T // 8)], bases[b * L:(b + 1) * L])
                           for b, w in enumerate(widths)])
    assert checker.mismatches(ty, "undelta_pack", 0, dcol, to_dev(want), n, aux=to_dev(bases), widths=dw_, offsets=doff) == 0
    assert checker.mismatches(ty, "undelta_pack", 0, dcol, corrupt(to_dev(want), at), n, aux=to_dev(bases), widths=dw_, offsets=doff) == 1


class BackgroundLoad:
    """Keeps every CU busy on a SECOND stream while the kernels under test run on the current one: a queue of large decode
    launches (u32 W=20, 2 M blocks = 13 GB of traffic, ~2 ms each) refilled before every call under test."""

    def __init__(self, fl):
        import torch
        self.torch, self.fl = torch, fl
        self.stream = torch.cuda.Stream()
        n = 2_000_000
        self.pk = torch.empty(n * 640, dtype=torch.uint32, device="cuda:0")
        rc = fl.load().fl_fill_random(self.pk.data_ptr(), self.pk.numel() * 4, 3, None)
        assert rc == 0, (rc, fl.load().fl_last_hip_error())
        self.out = torch.empty(n * 1024, dtype=torch.uint32, device="cuda:0")
        torch.cuda.synchronize()

    def refill(self, launches=3):
        with self.torch.cuda.stream(self.stream):
            for _ in range(launches):
                self.fl.BitPacking.unpack(20, self.pk, output=self.out)

    def drain(self):
        self.stream.synchronize()


FULL_WIDTHS = {"u8": (3, 5, 8), "u16": (3, 9, 13), "u32": (7, 12, 20, 31), "u64": (4, 17, 40)}


@pytest.mark.parametrize("policy", [0, 1, 2])
@pytest.mark.parametrize("ty", TYS)
def test_every_element_of_every_family_under_load(fl, checker, ty, policy):
    import torch
    T, L = tbits(ty), lanes(ty)
    tdt = getattr(torch, str(np.dtype(TYPES[ty][0])))
    n = 500_000 + 37                       # ragged against every tile size
    lib = fl.load()
    load = BackgroundLoad(fl)

    def filled(n_elems, seed):
        t = torch.empty(n_elems, dtype=tdt, device="cuda:0")
        nb = (n_elems * (T // 8)) & ~7
        assert lib.fl_fill_random(t.data_ptr(), nb, seed, None) == 0
        return t

    def under_load(call):
        load.refill()
        out = call()
        torch.cuda.current_stream().synchronize()
        return out

    un, bases, refs = filled(n * 1024, 5), filled(n * L, 6), filled(n, 7)
    lib.fl_internal_set_kernel_policy(policy)
    try:
        for op, call in (("delta", lambda: fl.Delta.delta(un, bases)), ("undelta", lambda: fl.Delta.undelta(un, bases)),
                         ("transpose", lambda: fl.Transpose.transpose(un)), ("untranspose", lambda: fl.Transpose.untranspose(un))):
            got = under_load(call)
            differing = checker.mismatches(ty, op, 0, un, got, n, aux=bases)
            assert differing == 0, (ty, policy, op, f"{differing} elements differ")
            del got
        mins, maxs = under_load(lambda: fl.BitPacking.block_min_max(un))
        differing = checker.mismatches(ty, "min_max", 0, un, mins, n, got2=maxs)
        assert differing == 0, (ty, policy, "block_min_max", f"{differing} elements differ")
        for w in FULL_WIDTHS[ty]:
            pk = filled(n * packed_len(ty, w), 11 + w)
            for op, src, call, kw in (
                    ("unpack", pk, lambda: fl.BitPacking.unpack(w, pk), {}),
                    ("unpack", pk, lambda: fl.FoR.unfor_pack(w, pk, refs), dict(aux=refs, aux_stride=1)),
                    ("unpack", pk, lambda: fl.FoR.unfor_pack(w, pk, refs[:1]), dict(aux=refs, aux_stride=0)),
                    ("pack", un, lambda: fl.BitPacking.pack(w, un), {}),
                    ("pack", un, lambda: fl.FoR.for_pack(w, un, refs), dict(aux=refs, aux_stride=1)),
                    ("undelta_pack", pk, lambda: fl.Delta.undelta_pack(w, pk, bases), dict(aux=bases)),
                    ("undelta_pack_untranspose", pk, lambda: fl.Delta.undelta_pack_untranspose(w, pk, bases), dict(aux=bases)),
                    ("transpose_delta_pack", un, lambda: fl.Delta.transpose_delta_pack(w, un, bases), dict(aux=bases)),
                    ("block_sums", pk, lambda: fl.BitPacking.unpack_block_sums(w, pk), {})):
                got = under_load(call)
                differing = checker.mismatches(ty, op, w, src, got, n, **kw)
                assert differing == 0, (ty, policy, w, op, kw.get("aux_stride"), f"{differing} elements differ")
                del got
            kc = (1 << w) // 3
            for name in ("<=", "==", ">"):
                mask = under_load(lambda: fl.BitPacking.unpack_compare(w, pk, name, kc))
                differing = checker.mismatches(ty, "compare", w, pk, mask, n, cmp_op=name, cmp_k=kc)
                assert differing == 0, (ty, policy, w, "compare", name, f"{differing} elements differ")
            del pk
        # mixed widths (always the wave-per-block kernels): seeded-random widths 0..T, offsets built on the device
        g = torch.Generator(device="cuda:0")
        g.manual_seed(900 + T)
        widths = torch.randint(0, T + 1, (n,), dtype=torch.int64, device="cuda:0", generator=g).to(torch.uint8)
        offsets, total = fl.widths_to_offsets(ty, widths)
        col = filled(max(int(total.item()) // (T // 8), 16), 13)
        got = under_load(lambda: fl.unpack_widths(widths, offsets, col, check=True))
        differing = checker.mismatches(ty, "unpack", 0, col, got, n, widths=widths, offsets=offsets)
        assert differing == 0, (ty, policy, "unpack_widths", f"{differing} elements differ")
        back = torch.zeros_like(col)
        under_load(lambda: fl.pack_widths(widths, offsets, un, back, check=True))
        differing = checker.mismatches(ty, "pack", 0, un, back, n, widths=widths, offsets=offsets)
        assert differing == 0, (ty, policy, "pack_widths", f"{differing} elements differ")
        # ... with FoR's and Delta's bodies (fl_<ty>_unfor_pack_widths, ..): the same checker ops with per-block widths
        for name, op, src, call, kw in (
                ("unfor_pack_widths", "unpack", col, lambda: fl.unfor_pack_widths(widths, offsets, col, refs), dict(aux=refs, aux_stride=1)),
                ("undelta_pack_widths", "undelta_pack", col, lambda: fl.undelta_pack_widths(widths, offsets, col, bases), dict(aux=bases)),
                ("undelta_pack_untranspose_widths", "undelta_pack_untranspose", col,
                 lambda: fl.undelta_pack_widths(widths, offsets, col, bases, untranspose=True), dict(aux=bases))):
            got = under_load(call)
            differing = checker.mismatches(ty, op, 0, src, got, n, widths=widths, offsets=offsets, **kw)
            assert differing == 0, (ty, policy, name, f"{differing} elements differ")
            del got
        for name, op, call, kw in (
                ("for_pack_widths", "pack", lambda o: fl.for_pack_widths(widths, offsets, un, refs, o), dict(aux=refs, aux_stride=1)),
                ("transpose_delta_pack_widths", "transpose_delta_pack", lambda o: fl.transpose_delta_pack_widths(widths, offsets, un, bases, o),
                 dict(aux=bases))):
            back.zero_()
            under_load(lambda: call(back))
            differing = checker.mismatches(ty, op, 0, un, back, n, widths=widths, offsets=offsets, **kw)
            assert differing == 0, (ty, policy, name, f"{differing} elements differ")
        load.drain()
    finally:
        lib.fl_internal_set_kernel_policy(0)
        torch.cuda.synchronize()
