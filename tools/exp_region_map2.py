#!/usr/bin/env python3
"""Round 3 experiment, continued: (1) classify every 8-GiB granule of one big allocation by the interference test of
tools/exp_region_map.py (a thin write stream next to a bulk read stream is slow iff both lie in granules of the same class);
(2) with that map in hand, unpack u32 W=7 (10 M blocks: 8.3 GiB read, 38 GiB written) with the input at offset 0 and the output
at every 8-GiB step -- which classes should a BULK write stream avoid or cover?  (3) the same for pack u32 W=7.
    python tools/exp_region_map2.py [size_gib=240]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402

dev = torch.device("cuda", 0)
lib = fl.load()
GiB = 1 << 30
size = int(sys.argv[1]) if len(sys.argv) > 1 else 240
slab = torch.empty(size * GiB, dtype=torch.uint8, device=dev)
assert lib.fl_fill_random(slab.data_ptr(), slab.numel(), 7, None) == 0


def timed(f, total, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    return total / sorted(ms)[len(ms) // 2] / 1e6


# ---- (1) classes ------------------------------------------------------------------------------------------------
w, n = 20, 2_500_000
ib, ob = n * 128 * w, n * 128
cmp_fn = lib.fl_u32_unpack_compare
kc = fl._lib.CTYPE["u32"]((1 << w) // 2)
G = size // 8


def probe(gi, gm):
    io, oo = gi * 8 * GiB, gm * 8 * GiB + 7 * GiB
    src, dst = slab[io:io + ib], slab[oo:oo + ob]
    return timed(lambda: cmp_fn(w, src.data_ptr(), 2, kc, n, dst.data_ptr(), None), ib + ob)


cls = [None] * G
names = "ABCDEFGH"
for c in range(len(names)):
    rep = next((g for g in range(G) if cls[g] is None), None)
    if rep is None:
        break
    cls[rep] = names[c]
    rates = {g: probe(rep, g) for g in range(G) if cls[g] is None}
    if not rates:
        break
    cut = (max(rates.values()) + min(rates.values())) / 2
    if max(rates.values()) - min(rates.values()) > 400:
        for g, r in rates.items():
            if r < cut:
                cls[g] = names[c]
cmap = "".join(x or "?" for x in cls)
print(f"allocation {size} GiB, class of every 8-GiB granule (thin-write interference test): {cmap}", flush=True)

# ---- (2), (3) bulk streams -------------------------------------------------------------------------------------
nb = 10_000_000
pb, ub = nb * 128 * 7, nb * 4096
for op in ("unpack", "pack"):
    in_b, out_b = (pb, ub) if op == "unpack" else (ub, pb)
    print(f"{op} u32 W=7, 10 M blocks, input at 0 (granules {cmap[:(in_b + 8 * GiB - 1) // (8 * GiB)]}); output at <offset GiB>: GB/s  [granules it covers]")
    start = (in_b + 8 * GiB - 1) // (8 * GiB) * 8
    for off in range(start, size - (out_b + GiB - 1) // GiB, 8):
        src = slab[:in_b].view(torch.uint32)
        dst = slab[off * GiB:off * GiB + out_b].view(torch.uint32)
        f = (lambda: fl.BitPacking.unpack(7, src, output=dst)) if op == "unpack" else (lambda: fl.BitPacking.pack(7, src, output=dst))
        r = timed(f, in_b + out_b)
        covered = cmap[off // 8:(off * GiB + out_b + 8 * GiB - 1) // (8 * GiB)]
        print(f"  out@{off:3d}: {r:6.0f} ({r / 8000:.3f})  [{covered}]", flush=True)
    if op == "unpack":
        assert lib.fl_fill_random(slab.data_ptr(), ub, 9, None) == 0      # pack's input: full-entropy values
