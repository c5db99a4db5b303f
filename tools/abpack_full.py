#!/usr/bin/env python3
"""A/B on the GPU box at BASELINE-like sizes: pack / unpack through the cell-column kernels vs the wave-per-block kernels at
3/4/5/6/8 waves per SIMD, on the SAME buffers, placed like bench.py places them (zone-aware: fastlanes_amd/placement.py),
full-entropy inputs.
    FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so python tools/abpack_full.py [--all] [--gb 64] [--rounds 3]
Default: BASELINE's configs 2, 3 (and u32 W=12) at the full 10 M blocks.  --all: every (T, W) at min(10 M blocks, --gb GB per
launch), printed in tools/abuniform's row format -- the input of tools/make_dispatch.py (which kernel and which occupancy
stream faster moves with the column size and the allocation: the 16-GiB hipMalloc sweeps of abuniform prefer fewer waves than
a 50-100 GB column does)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402
from fastlanes_amd import placement as pl  # noqa: E402

lib = fl.load()
assert b"FULL build" in lib.fl_version(), "run against libfastlanes_amd_full.so (FL_LIB=...): policy 1 must mean the cell-column kernel"
dev = torch.device("cuda", 0)
ALL = "--all" in sys.argv
GB = float(sys.argv[sys.argv.index("--gb") + 1]) if "--gb" in sys.argv else 64.0
ROUNDS = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else (3 if ALL else 5)
WAVES = (3, 4, 5, 6, 8)
TD = {"u8": (torch.uint8, 8), "u16": (torch.uint16, 16), "u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
cases = [("u32", 7), ("u64", 17), ("u32", 12)]
if ALL:
    cases = [(ty, w) for ty in ("u32", "u64", "u16", "u8") for w in range(TD[ty][1] + 1)]
if "--types" in sys.argv:
    cases = [c for c in cases if c[0] in sys.argv[sys.argv.index("--types") + 1].split(",")]
print(f"GB/s (algorithmic bytes), median of {ROUNDS}, min(10 M blocks, {GB:.0f} GB) per launch, zone-aware placement; cc = cell-column kernel, "
      "wpb = wave-per-block at 3 4 5 6 8 waves/SIMD")
def fill(t, seed):
    if t.numel() & ~7:
        assert lib.fl_fill_random(t.data_ptr(), t.numel() & ~7, seed, None) == 0


for ty, W in cases:
    tdt, T = TD[ty]
    esz = T // 8
    bpb = 128 * W + 1024 * esz
    n = min(10_000_000, int(GB * 1e9 / bpb))
    # zone-aware placement (fastlanes_amd/placement.py), one slab per direction: the input inside one 64-GiB zone, the output
    # straddling two -- the layout bench.py uses, so that a row compares the two designs where the bench measures them
    if "--constructed" in sys.argv:
        # second half of round 6: each direction's buffers in a CONSTRUCTED pair of exactly the row's size (fl_column_pair_alloc:
        # FL_LAYOUT_INTERLEAVED; in memory of one class every variant reads the same figure).  The library keeps the chunks between rows;
        # one process per element type (--types): a pair's address ranges are never re-used.
        lib.fl_internal_pair_chunk_cache(128)
        slab_u = pl.ColumnPair(max(n * 128 * W, 256), n * 1024 * esz, dev, layout="interleaved")
        slab_p = pl.ColumnPair(n * 1024 * esz, max(n * 128 * W, 256), dev, layout="interleaved")
        pk8, un8, uni8, pko8 = slab_u.input[:n * 128 * W], slab_u.output, slab_p.input, slab_p.output[:n * 128 * W]
        print(f"# u{T} W={W}: constructed pairs, measured classes (input first): decode {slab_u.classes} | encode {slab_p.classes}", flush=True)
    else:
        slab_u, pk8, _, un8 = pl.column_pair(n * 128 * W, n * 1024 * esz, dev)        # unpack: packed in, unpacked out
        slab_p, uni8, _, pko8 = pl.column_pair(n * 1024 * esz, n * 128 * W, dev)      # pack: full-entropy values in (pack truncates)
    fill(pk8, 2)
    pk_in, un_out = pk8.view(tdt), un8.view(tdt)
    fill(uni8, 1)
    un, pk_out = uni8.view(tdt), pko8.view(tdt)
    row = {}
    for name, f in (("unpack", lambda: fl.BitPacking.unpack(W, pk_in, output=un_out, n_blocks=n)), ("pack", lambda: fl.BitPacking.pack(W, un, output=pk_out))):
        res = {}
        pols = [1] + [2 + 256 * w for w in WAVES]
        for _ in range(ROUNDS):
            for p in pols:
                lib.fl_internal_set_kernel_policy(p)
                f(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); torch.cuda.synchronize()
                res.setdefault(p, []).append(a.elapsed_time(b))
        row[name] = [n * bpb / sorted(res[p])[len(res[p]) // 2] / 1e6 for p in pols]
    lib.fl_internal_set_kernel_policy(0)
    u, p = row["unpack"], row["pack"]
    print(f"u{T:<2d} W={W:<2d} | unpack cc {u[0]:6.0f}  wpb " + " ".join(f"{x:6.0f}" for x in u[1:]) +
          f" | pack cc {p[0]:6.0f}  wpb " + " ".join(f"{x:6.0f}" for x in p[1:]), flush=True)
    del pk_in, pk_out, un, un_out, pk8, un8, uni8, pko8
    if "--constructed" in sys.argv:
        slab_u.free(); slab_p.free()
    del slab_u, slab_p
    torch.cuda.empty_cache()
