#!/usr/bin/env python3
"""A/B of kernel POLICIES (fastlanes_amd_internal.h: fl_internal_set_kernel_policy) for uniform-width calls on the same buffers,
launches interleaved round-robin:  python tools/abpolicy.py <op,op,..> <type:width,..> <policy,policy,..> [--gb 8] [--reps 9]
    ops       unpack pack unfor_pack for_pack undelta_pack
    policy    0 = the library's own choice; 1 = cell-column kernels; 2 + 256 * waves = wave-per-block kernel at `waves` per SIMD
e.g. abpolicy.py unfor_pack,unpack u64:18,u32:9 0,0x302,0x402,0x502,0x602,0x802"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402
from bench import rand_u8  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("ops")
ap.add_argument("cases")
ap.add_argument("policies")
ap.add_argument("--gb", type=float, default=8.0)
ap.add_argument("--reps", type=int, default=9)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = fl.load()
TDT = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
ESZ = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}
pols = [int(x, 0) for x in args.policies.split(",")]


def name(p):
    return "auto" if p == 0 else "cell-column" if p == 1 else f"wave x{(p >> 8) & 255}" if (p & 255) == 2 else hex(p)


print("GB/s of algorithmic bytes (fraction of 8 TB/s), median of %d round-robin launches; policies: %s" % (args.reps, "  ".join(name(p) for p in pols)))
for case in args.cases.split(","):
    ty, w = case.split(":")[0], int(case.split(":")[1])
    T, esz = ESZ[ty] * 8, ESZ[ty]
    for op in args.ops.split(","):
        bpb = 128 * w + 128 * T + (128 if op == "undelta_pack" else 0)
        n = int(args.gb * 1e9 / bpb)
        pk = rand_u8(n * 128 * w, 2, dev).view(TDT[ty])
        un = rand_u8(n * 128 * T, 3, dev).view(TDT[ty])
        aux = rand_u8(n * 128, 4, dev).view(TDT[ty])
        refs = aux[:n]
        f = {"unpack": lambda: fl.BitPacking.unpack(w, pk, output=un), "pack": lambda: fl.BitPacking.pack(w, un, output=pk),
             "unfor_pack": lambda: fl.FoR.unfor_pack(w, pk, refs, output=un), "for_pack": lambda: fl.FoR.for_pack(w, un, refs, output=pk),
             "undelta_pack": lambda: fl.Delta.undelta_pack(w, pk, aux, output=un)}[op]
        ms = {p: [] for p in pols}
        for p in pols:
            lib.fl_internal_set_kernel_policy(p)
            f()
        torch.cuda.synchronize()
        for _ in range(args.reps):
            for p in pols:
                lib.fl_internal_set_kernel_policy(p)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                ms[p].append(a.elapsed_time(b))
        lib.fl_internal_set_kernel_policy(0)
        print(f"{op:13s} {ty:4s} W={w:<2d} n={n:>8d} | " + "  ".join(f"{n * bpb / sorted(ms[p])[len(ms[p]) // 2] / 8e9:.3f}" for p in pols), flush=True)
        del pk, un, aux, refs
        torch.cuda.empty_cache()
