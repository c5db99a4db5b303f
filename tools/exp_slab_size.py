#!/usr/bin/env python3
"""Round 3 experiment: does the SIZE of the allocation the zone-aware layout is carved from matter?  placement.column_pair() allocates
just enough (output centred on the 64-GiB multiple: 64 GiB + half the output), so the part behind the 64-GiB multiple is a small
remainder the driver may back from anywhere.  Same layout (input at 0, output centred on 64 GiB) inside allocations of growing size,
fresh allocation each (torch.cuda.empty_cache() in between), unpack u32 W=7 (10 M blocks) and unpack_compare u32 W=20 / u64 W=17.
    python tools/exp_slab_size.py"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402

dev = torch.device("cuda", 0)
lib = fl.load()
GiB = 1 << 30
n = 10_000_000


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


cases = [("unpack", "u32", 7), ("compare", "u32", 20), ("compare", "u64", 17), ("compare", "u32", 7)]
TD = {"u32": (torch.uint32, 4), "u64": (torch.uint64, 8)}
print("GB/s; same layout (input at offset 0, output centred on 64 GiB) inside one fresh allocation of the size named")
for size in (0, 96, 128, 140, 200):
    row = []
    for op, ty, w in cases:
        tdt, esz = TD[ty]
        ib = n * 128 * w
        ob = n * 1024 * esz if op == "unpack" else n * 128
        out_off = (64 * GiB - ob // 2) & ~255
        need = out_off + ob
        total = max(need, size * GiB)
        torch.cuda.empty_cache()
        slab = torch.empty(total, dtype=torch.uint8, device=dev)
        assert lib.fl_fill_random(slab.data_ptr(), ib & ~7, 7, None) == 0
        src, dst = slab[:ib].view(tdt), slab[out_off:out_off + ob]
        if op == "unpack":
            d = dst.view(tdt)
            r = timed(lambda: fl.BitPacking.unpack(w, src, output=d), ib + ob)
        else:
            m = dst.view(torch.int32)
            r = timed(lambda: fl.BitPacking.unpack_compare(w, src, "<", (1 << w) // 2, n_blocks=n, output=m), ib + ob)
        row.append(f"{op} {ty} W={w}: {r:6.0f} ({total / GiB:5.1f} GiB)")
        del slab, src, dst
        d = m = None
        gc.collect()
    print(("just enough" if size == 0 else f"{size:3d} GiB    ") + " | " + " | ".join(row), flush=True)
