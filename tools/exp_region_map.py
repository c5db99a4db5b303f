#!/usr/bin/env python3
"""Round 3 experiment: the interference map of a thin WRITE stream and a bulk READ stream over (almost) the whole device memory.
One allocation of <size> GiB; unpack_compare u32 W=20 on 2.5 M blocks (6 GiB in, 0.3 GiB mask) with the input at every 16-GiB
step and the mask at every 8-GiB step (skipping overlaps).  Rows = input offset, columns = mask offset, cell = GB/s / 100.
    python tools/exp_region_map.py [size_gib=240]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402

dev = torch.device("cuda", 0)
lib = fl.load()
GiB = 1 << 30
size = int(sys.argv[1]) if len(sys.argv) > 1 else 240
w, n = 20, 2_500_000
ib, ob = n * 128 * w, n * 128
slab = torch.empty(size * GiB, dtype=torch.uint8, device=dev)
assert lib.fl_fill_random(slab.data_ptr(), slab.numel(), 7, None) == 0
fn = lib.fl_u32_unpack_compare
k = fl._lib.CTYPE["u32"]((1 << w) // 2)


def rate(io, oo):
    src, dst = slab[io:io + ib], slab[oo:oo + ob]
    for _ in range(2):
        assert fn(w, src.data_ptr(), 2, k, n, dst.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(w, src.data_ptr(), 2, k, n, dst.data_ptr(), None); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    return (ib + ob) / sorted(ms)[2] / 1e6


cols = list(range(0, size - 1, 8))
print(f"unpack_compare u32 W={w}, {n} blocks; allocation {size} GiB; cell = GB/s / 100 ('--' = would overlap)")
print("in\\mask " + " ".join(f"{c:3d}" for c in cols))
for i in range(0, size - 8, 16):
    cells = []
    for c in cols:
        io, oo = i * GiB, c * GiB + 7 * GiB          # the mask in the last GiB of its 8-GiB window: away from an input in the same window
        if oo + ob > size * GiB or not (io + ib <= oo or oo + ob <= io):
            cells.append(" --")
        else:
            cells.append(f"{rate(io, oo) / 100:3.0f}")
    print(f"{i:7d} " + " ".join(cells), flush=True)
