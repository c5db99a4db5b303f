#!/usr/bin/env python3
"""Workload for the round-3 SQ / LDS counter passes: the kernels VERDICT r02 lists furthest below the roofline -- the fused
consumers (unpack_compare, unpack_block_sums) and the narrow types (pack u16 W=3, undelta_pack u16 W=9, u8 mixed widths).
Run under rocprofv3 --kernel-trace --pmc <counters> (tools/gpu/sq_counters.sh tools/pmc_probe_r03.py gpurun_out/r03)."""
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


def packed(ty, w, nbytes_per_block):
    n = int(GB * 1e9 / nbytes_per_block)
    return n, rand_u8(n * 128 * w, 11, dev).view(TDT[ty])


for ty, w in (("u16", 3), ("u32", 7), ("u32", 20), ("u8", 3), ("u64", 17)):
    n, pk = packed(ty, w, 128 * w + 128)
    for _ in range(3):
        fl.BitPacking.unpack_compare(w, pk, "<", (1 << w) // 2)
    torch.cuda.synchronize()
    for _ in range(3):
        fl.BitPacking.unpack_block_sums(w, pk)
    torch.cuda.synchronize()
    del pk
for ty, w in (("u16", 3), ("u8", 3)):
    T = ESZ[ty] * 8
    n = int(GB * 1e9 / (128 * w + 128 * T))
    un = rand_u8(n * 128 * T, 12, dev).view(TDT[ty])
    out = torch.empty(n * 128 * w // ESZ[ty], dtype=TDT[ty], device=dev)
    for _ in range(3):
        fl.BitPacking.pack(w, un, output=out)
    torch.cuda.synchronize()
    del un, out
n, pk = packed("u16", 9, 128 * 9 + 128 + 2048)
bases = rand_u8(n * 128, 13, dev).view(torch.uint16)
out = torch.empty(n * 1024, dtype=torch.uint16, device=dev)
for _ in range(3):
    fl.Delta.undelta_pack(9, pk, bases, output=out)
torch.cuda.synchronize()
del pk, bases, out
# u8 mixed widths 1..8
n = int(GB * 1e9 / (128 * 4.5 + 1024))
g = torch.Generator(device=dev)
g.manual_seed(5)
widths = torch.randint(1, 9, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.uint8)
offsets, total = fl.widths_to_offsets("u8", widths)
col = rand_u8(int(total.item()), 14, dev)
out = torch.empty(n * 1024, dtype=torch.uint8, device=dev)
for _ in range(3):
    fl.unpack_widths(widths, offsets, col, output=out, check=False)
back = torch.empty_like(col)
for _ in range(3):
    fl.pack_widths(widths, offsets, out, back, check=False)
torch.cuda.synchronize()
print("pmc_probe_r03 done")
