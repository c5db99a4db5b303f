#!/usr/bin/env python3
"""Round 3 experiment: how far from the bulk stream must a THIN stream live?  (a) unpack_compare: the 128-byte-per-block mask written
at a growing gap behind (or in front of) the packed input; (b) undelta_pack: the 128-byte-per-block bases READ at a growing gap
behind the packed input, output centred on the 64-GiB multiple as bench.py places it.  One allocation, 10 M blocks.
    python tools/exp_thin_stream.py <compare|undelta_pack> <ty> <width>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402

dev = torch.device("cuda", 0)
lib = fl.load()
op, ty, w = sys.argv[1], sys.argv[2], int(sys.argv[3])
n = 10_000_000
TD = {"u8": (torch.uint8, 1), "u16": (torch.uint16, 2), "u32": (torch.uint32, 4), "u64": (torch.uint64, 8)}
tdt, esz = TD[ty]
GiB = 1 << 30
ib, tb, ub = n * 128 * w, n * 128, n * 1024 * esz
slab = torch.empty(150 * GiB, dtype=torch.uint8, device=dev)
assert lib.fl_fill_random(slab.data_ptr(), slab.numel(), 7, None) == 0


def timed(f, total):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    return total / sorted(ms)[3] / 1e6


def at(off_gib, nbytes):
    o = int(off_gib * GiB) & ~255
    return slab[o:o + nbytes]


ig = ib / GiB
rows = []
if op == "compare":
    fn = getattr(lib, f"fl_{ty}_unpack_compare")
    k = fl._lib.CTYPE[ty]((1 << w) // 2)
    base = 16.0                                             # room for a mask in front of the input
    src = at(base, ib).view(tdt)
    for name, off in [("mask 8 GiB in front", base - 8), ("2 GiB in front", base - 2), ("directly in front", base - tb / GiB - 0.001)] + \
                     [(f"gap {g:g} GiB behind", base + ig + g) for g in (0.001, 0.25, 1, 2, 4, 8, 16, 24)]:
        dst = at(off, tb)
        rows.append((name, timed(lambda: fn(w, src.data_ptr(), 2, k, n, dst.data_ptr(), None), ib + tb)))
else:
    out_off = 64.0 - ub / GiB / 2
    src, dst = at(0, ib).view(tdt), at(out_off, ub).view(tdt)
    for g in (0.001, 0.25, 1, 2, 4, 8, 16):
        if ig + g + tb / GiB > out_off:
            break
        bases = at(ig + g, tb).view(tdt)
        rows.append((f"bases gap {g:g} GiB behind the input", timed(lambda: fl.Delta.undelta_pack(w, src, bases, output=dst), ib + tb + ub)))
    bases = at(out_off + ub / GiB + 1.0, tb).view(tdt)
    rows.append(("bases 1 GiB behind the OUTPUT", timed(lambda: fl.Delta.undelta_pack(w, src, bases, output=dst), ib + tb + ub)))
    bases = at(130.0, tb).view(tdt)
    rows.append(("bases @130 GiB", timed(lambda: fl.Delta.undelta_pack(w, src, bases, output=dst), ib + tb + ub)))
print(f"{op} {ty} W={w}, 10 M blocks (bulk input {ig:.1f} GiB, thin stream {tb / GiB:.1f} GiB):")
for k_, v in rows:
    print(f"  {k_:40s} {v:6.0f} GB/s ({v / 8000:.3f})", flush=True)
