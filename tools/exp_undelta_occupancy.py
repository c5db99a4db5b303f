#!/usr/bin/env python3
"""VERDICT r05 "next" #4: undelta_pack u32 / u64 next to unpack of the same column in a CONSTRUCTED layout (fl_column_pair_alloc:
FL_LAYOUT_INTERLEAVED, so that placement is not the bound), at every occupancy the wave-per-block kernel can be launched with
(fl_internal_set_kernel_policy: 2 + 256 * waves) and as the dispatch table launches it.  Fractions of the 8 TB/s."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402
from fastlanes_amd import placement as pl  # noqa: E402

lib = fl.load()
dev = torch.device("cuda:0")
TD = {"u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
cases = [("u32", 7), ("u32", 12), ("u32", 20), ("u32", 28), ("u64", 9), ("u64", 17), ("u64", 33), ("u64", 50)]
if len(sys.argv) > 1:
    cases = [(c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[1].split(",")]
print("type W | unpack (table) | undelta_pack: table | wave-per-block at 3 4 5 6 7 8 waves/SIMD | cell-column   (fraction of 8 TB/s; constructed layout)")
for ty, W in cases:
    tdt, T = TD[ty]
    n = min(10_000_000, int(40e9 / (128 * W + 128 * T + 128)))
    pair = pl.ColumnPair(n * 128 * W, n * 128 * T, dev, aux_bytes=n * 128, layout="interleaved")
    assert lib.fl_fill_random(pair.input.data_ptr(), (n * 128 * W) & ~7, 3, None) == 0 and lib.fl_fill_random(pair.aux.data_ptr(), n * 128, 4, None) == 0
    pk, bases, out = pair.input.view(tdt), pair.aux.view(tdt), pair.output.view(tdt)
    rows = [("unpack", 0, lambda: fl.BitPacking.unpack(W, pk, output=out), n * (128 * W + 128 * T)),
            ("table", 0, lambda: fl.Delta.undelta_pack(W, pk, bases, output=out), n * (128 * W + 128 + 128 * T))]
    rows += [(f"w{k}", 2 + 256 * k, rows[1][2], rows[1][3]) for k in (3, 4, 5, 6, 7, 8)] + [("cc", 1, rows[1][2], rows[1][3])]
    res = {r[0]: [] for r in rows}
    for rnd in range(5):
        for name, pol, f, nbytes in rows:
            lib.fl_internal_set_kernel_policy(pol)
            ms = []
            for i in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                if i:
                    ms.append(a.elapsed_time(b))
            res[name].append(nbytes / statistics.median(ms) / 8e9)
    lib.fl_internal_set_kernel_policy(0)
    m = {k: statistics.median(v) for k, v in res.items()}
    print(f"{ty} W={W:<2d} n={n} | {m['unpack']:.3f} | {m['table']:.3f} | " + " ".join(f"{m['w%d' % k]:.3f}" for k in (3, 4, 5, 6, 7, 8)) + f" | {m['cc']:.3f}   classes {pair.classes}", flush=True)
    pair.free()
