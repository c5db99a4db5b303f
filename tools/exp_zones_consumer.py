#!/usr/bin/env python3
"""Round 3 experiment: where should a read-dominated consumer's small output live?  unpack_compare at <n_blocks> blocks with the packed
input at offset 0 of one allocation and the mask at chosen offsets (the device memory behaves as 64-GiB zones: DESIGN.md 4), plus the
input moved so that IT straddles a zone boundary.
    python tools/exp_zones_consumer.py <ty> <width> [n_blocks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402

dev = torch.device("cuda", 0)
lib = fl.load()
ty, w = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
TD = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
tdt = TD[ty]
ib, ob = n * 128 * w, n * 128
GiB = 1 << 30
slab = torch.empty(140 * GiB, dtype=torch.uint8, device=dev)
assert lib.fl_fill_random(slab.data_ptr(), slab.numel(), 7, None) == 0


def rate(in_off, out_off):
    io, oo = int(in_off * GiB) & ~255, int(out_off * GiB) & ~255
    assert io + ib <= oo or oo + ob <= io, "overlap"
    src, dst = slab[io:io + ib].view(tdt), slab[oo:oo + ob]
    fn = getattr(lib, f"fl_{ty}_unpack_compare")
    k = fl._lib.CTYPE[ty]((1 << w) // 2)

    def f():
        assert fn(w, src.data_ptr(), 2, k, n, dst.data_ptr(), None) == 0      # 2 = FL_CMP_LT
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    return (ib + ob) / sorted(ms)[3] / 1e6


ig, og = ib / GiB, ob / GiB
layouts = [("in@0, mask right behind it", 0.0, ig + 0.01), ("in@0, mask@32", 0.0, max(32.0, ig + 0.01)), ("in@0, mask ends at 64", 0.0, 64.0 - og),
           ("in@0, mask centred on 64", 0.0, 64.0 - og / 2), ("in@0, mask@64", 0.0, 64.0), ("in@0, mask@100", 0.0, 100.0),
           ("in centred on 64, mask@0", 64.0 - ig / 2, 0.0), ("in@64, mask@0", 64.0, 0.0), ("in@64, mask right behind it", 64.0, 64.0 + ig + 0.01)]
res = []
for name, i, o in layouts:
    if i == 0.0 and o < ig:
        continue
    if o == 0.0 and og > i:
        continue
    res.append((name, rate(i, o)))
print(f"unpack_compare {ty} W={w} n={n} (input {ig:.1f} GiB, mask {og:.1f} GiB):")
for k, v in res:
    print(f"  {k:32s} {v:6.0f} GB/s ({v / 8000:.3f})", flush=True)
