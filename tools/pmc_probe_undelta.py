#!/usr/bin/env python3
"""Workload for the SQ counter passes over undelta_pack next to unpack (VERDICT r05 "next" #4): u32 W=12 / 20, u64 W=17 / 33, uniform
widths, the kernels the dispatch table launches.  bash tools/gpu/sq_counters.sh tools/pmc_probe_undelta.py gpurun_out/<dir>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402
from bench import rand_u8  # noqa: E402

dev = torch.device("cuda:0")
TD = {"u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
for ty, W in (("u32", 12), ("u32", 20), ("u64", 17), ("u64", 33)):
    tdt, T = TD[ty]
    n = int(8e9 / (128 * W + 128 * T))
    pk = rand_u8(n * 128 * W, 2, dev).view(tdt)
    bases = rand_u8(n * 128, 3, dev).view(tdt)
    out = torch.empty(n * 1024, dtype=tdt, device=dev)
    for _ in range(3):
        fl.BitPacking.unpack(W, pk, output=out)
    for _ in range(3):
        fl.Delta.undelta_pack(W, pk, bases, output=out)
    torch.cuda.synchronize()
    del pk, bases, out
print("pmc_probe_undelta done")
