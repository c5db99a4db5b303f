#!/usr/bin/env python3
"""A/B on the GPU box at BASELINE's full 10 M blocks: pack / unpack of configs 2 and 3 through the cell-column kernels vs the
wave-per-block kernels at several occupancies, same buffers, full-entropy inputs (fl_internal_set_kernel_policy)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402
from bench import rand_u8  # noqa: E402

lib = fl.load()
dev = torch.device("cuda", 0)
n = 10_000_000
WAVES = (3, 4, 5, 6, 8)
print("GB/s at 10 M blocks, median of 5; cc, then wave-per-block at", WAVES, "waves/SIMD; '*' = what the automatic policy runs")
for ty, tdt, T, W in (("u32", torch.uint32, 32, 7), ("u64", torch.uint64, 64, 17), ("u32", torch.uint32, 32, 12)):
    esz = T // 8
    un = rand_u8(n * 1024 * esz, 1, dev).view(tdt)
    pk = rand_u8(n * 128 * W, 2, dev).view(tdt)
    nbytes = n * (128 * W + 1024 * esz)
    for name, f in (("pack", lambda: fl.BitPacking.pack(W, un, output=pk)), ("unpack", lambda: fl.BitPacking.unpack(W, pk, output=un))):
        if name == "unpack":
            pk = rand_u8(n * 128 * W, 3, dev).view(tdt)
        res = {}
        pols = [0, 1] + [2 + 256 * w for w in WAVES]
        for _ in range(5):
            for p in pols:
                lib.fl_internal_set_kernel_policy(p)
                f(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); torch.cuda.synchronize()
                res.setdefault(p, []).append(a.elapsed_time(b))
        g = [nbytes / sorted(res[p])[2] / 1e6 for p in pols]
        print(f"{ty} W={W:<2d} {name:6s} | auto {g[0]:6.0f} | cc {g[1]:6.0f} | wpb " + " ".join(f"{x:6.0f}" for x in g[2:]), flush=True)
        if name == "pack":
            un = rand_u8(n * 1024 * esz, 4, dev).view(tdt)      # unpack overwrote nothing yet, but keep inputs fresh per op
    lib.fl_internal_set_kernel_policy(0)
    del un, pk
