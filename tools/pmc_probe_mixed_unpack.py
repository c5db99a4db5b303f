#!/usr/bin/env python3
"""Workload for tools/gpu/sq_wavelife.sh: unpack_widths / pack_widths over seeded-random mixed-width columns of every type, ~6 GB each
(where does a wavefront's life go -- the narrow types' decode sits 5 % behind the wide types' on the same bytes per wavefront)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402
from bench import rand_u8  # noqa: E402

dev = torch.device("cuda:0")
GB = float(os.environ.get("FL_GB", "6"))
TDT = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
ESZ = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}
for ty in os.environ.get("FL_TYPES", "u8,u16,u32,u64").split(","):
    T = ESZ[ty] * 8
    n = int(GB * 1e9 / (128 * T * 1.5))
    g = torch.Generator(device=dev)
    g.manual_seed(31 + T)
    widths = torch.randint(1, T, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.uint8)
    offsets, total = fl.widths_to_offsets(ty, widths)
    col = rand_u8(int(total.item()), 14, dev).view(TDT[ty])
    un = torch.empty(n * 1024, dtype=TDT[ty], device=dev)
    back = torch.empty_like(col)
    for _ in range(3):
        fl.unpack_widths(widths, offsets, col, output=un, check=False)
    for _ in range(3):
        fl.pack_widths(widths, offsets, un, back, check=False)
    torch.cuda.synchronize()
    del col, un, back
print("pmc_probe_mixed_unpack done")
