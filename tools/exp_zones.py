#!/usr/bin/env python3
"""Round 3 experiment: one codec call at 10 M blocks with (a) separately allocated torch tensors (bench.py's pattern until round 3)
and (b..) input and output carved from ONE torch allocation at chosen offsets (tools/abplacement2/3: 64-GiB parts of the device
memory behave as separate zones; traffic confined to one zone tops out near 6.2 TB/s, traffic over two zones reaches 6.8).
    python tools/exp_zones.py <ty> <width> <unpack|pack|undelta_pack>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fastlanes_amd as fl  # noqa: E402

dev = torch.device("cuda", 0)
lib = fl.load()
ty, w, op = (sys.argv[1], int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else ("u32", 7, "unpack")
TD = {"u8": (torch.uint8, 1), "u16": (torch.uint16, 2), "u32": (torch.uint32, 4), "u64": (torch.uint64, 8)}
tdt, esz = TD[ty]
n = 10_000_000
pb, ub, bb = n * 128 * w, n * 1024 * esz, (n * 128 if op == "undelta_pack" else 0)
ib, ob = (ub, pb) if op == "pack" else (pb, ub)
GiB = 1 << 30


def rate(src, dst, bases):
    if op == "unpack":
        f = lambda: fl.BitPacking.unpack(w, src, output=dst)
    elif op == "pack":
        f = lambda: fl.BitPacking.pack(w, src, output=dst)
    else:
        f = lambda: fl.Delta.undelta_pack(w, src, bases, output=dst)
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    return (ib + ob + bb) / sorted(ms)[3] / 1e6


def fill(t):
    if t.numel():
        assert lib.fl_fill_random(t.data_ptr(), (t.numel() * t.element_size()) & ~7, 7, None) == 0


res = []
src = torch.empty(ib, dtype=torch.uint8, device=dev); dst = torch.empty(ob, dtype=torch.uint8, device=dev); bas = torch.empty(bb, dtype=torch.uint8, device=dev)
fill(src); fill(bas)
res.append(("separate", rate(src.view(tdt), dst.view(tdt), bas.view(tdt))))
del src, dst, bas
torch.cuda.empty_cache()
slab = torch.empty(200 * GiB, dtype=torch.uint8, device=dev)
fill(slab)
big, small = (ob, ib + bb) if ob >= ib else (ib, ob)
layouts = {
    "in@0, out@64GiB": (0.0, 64.0),
    "big buffer centred on 64 GiB, small one @130": None,
    "big centred on 64, small @0 (ends before big starts?)": None,
    "in@0, out right behind it (same zone)": (0.0, (ib + bb) / GiB + 0.01),
}
for name, lay in layouts.items():
    if lay is None:
        c = 64.0 - big / GiB / 2
        so = 130.0 if "130" in name else 0.0
        if so == 0.0 and small / GiB > c:
            continue
        in_off, out_off = (so, c) if ob >= ib else (c, so)
    else:
        in_off, out_off = lay
    io, oo = int(in_off * GiB) & ~255, int(out_off * GiB) & ~255
    s = slab[io:io + ib].view(tdt)
    bs = slab[io + ib + 256 - (ib % 256 or 256):][:bb].view(tdt) if bb else None
    d = slab[oo:oo + ob].view(tdt)
    res.append((name, rate(s, d, bs)))
print(f"{op} {ty} W={w}: " + " | ".join(f"{k} {v:6.0f} ({v / 8000:.3f})" for k, v in res), flush=True)
