#!/usr/bin/env python3
"""Workload for the round-2 SQ / LDS counter passes: the wave-per-block kernels at their shipped shapes -- the mixed-width
column (config 5), u64 W=17 unpack / pack (config 3), u32 W=7 pack, u16 delta -- next to the cell-column headline kernel.
Run under rocprofv3 --kernel-trace --pmc <counters> (tools/gpu/r02_sq_counters.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402
from bench import Workload, rand_u8  # noqa: E402

dev = torch.device("cuda:0")
n = int(os.environ.get("FL_BLOCKS", "4000000"))
w = Workload("u32_mixed_unpack", n, 0, 0, dev)
for _ in range(3):
    w.step()
back = torch.empty_like(w.src)
for _ in range(3):
    fl.pack_widths(w.widths, w.offsets, w.dst, back, check=False)
torch.cuda.synchronize()
del w, back
for name in ("u32_w7_unpack", "u32_w7_pack", "u64_w17_unpack", "u64_w17_pack", "u32_w12_undelta_pack"):
    w = Workload(name, n, 0, 0, dev)
    for _ in range(3):
        w.step()
    torch.cuda.synchronize()
    del w
v = rand_u8(n * 2048, 5, dev).view(torch.uint16)
b = rand_u8(n * 128, 6, dev).view(torch.uint16)
o = torch.empty_like(v)
for _ in range(3):
    fl.Delta.delta(v, b, output=o)
    fl.Delta.undelta(v, b, output=o)
torch.cuda.synchronize()
print("pmc_probe_r02 done")
