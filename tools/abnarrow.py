#!/usr/bin/env python3
"""Uniform-width pack / unpack of the NARROW types: the table's kernel against the wave-per-block kernel with several blocks in
flight per wavefront (all requested up front by LDS-DMA -- what the mixed-width u8 kernels do), same buffers, round-robin.
    python tools/abnarrow.py [--reps 9] [--gb 16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402

lib = fl.load()
dev = torch.device("cuda:0")
TDT = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--gb", type=float, default=16.0)
ap.add_argument("--for", dest="ffor", action="store_true", help="FoR's bodies (for_pack / unfor_pack with a reference per block) instead of pack / unpack")
ap.add_argument("--wide", action="store_true", help="u32 W=7 / 20 and u64 W=17 instead of the narrow types (does the headline want more blocks in flight?)")
args = ap.parse_args()
POL = {"table": 0, "wpb 8x1": 2 + 256 * 8, "wpb 8x2 pf": 2 + 256 * 8 + 65536 * 2 + (1 << 24), "wpb 8x4 pf": 2 + 256 * 8 + 65536 * 4 + (1 << 24),
       "wpb 8x8 pf": 2 + 256 * 8 + 65536 * 8 + (1 << 24), "wpb 6x4 pf": 2 + 256 * 6 + 65536 * 4 + (1 << 24), "cell-column": 1}
print(f"# {lib.fl_version().decode()}\n# fraction of 8 TB/s (algorithmic bytes), median of {args.reps} round-robin launches; wpb AxB = wave-per-block kernel at A "
      "waves/SIMD, B blocks per wavefront, pf = all requested up front by LDS-DMA\n" + f"{'case':22s} " + " ".join(f"{k:>12s}" for k in POL), flush=True)
if args.wide:
    POL = {"table": 0, "wpb 8x1": 2 + 256 * 8, "wpb 6x1": 2 + 256 * 6, "wpb 8x2 pf": 2 + 256 * 8 + 65536 * 2 + (1 << 24), "wpb 5x2 pf": 2 + 256 * 5 + 65536 * 2 + (1 << 24),
           "wpb 4x2 pf": 2 + 256 * 4 + 65536 * 2 + (1 << 24), "wpb 3x3 pf": 2 + 256 * 3 + 65536 * 3 + (1 << 24), "wpb 8x2": 2 + 256 * 8 + 65536 * 2}
    print(f"{'case':22s} " + " ".join(f"{k:>12s}" for k in POL), flush=True)
for ty, T in ((("u32", 32), ("u64", 64)) if args.wide else (("u8", 8), ("u16", 16))):
    for w in ((7, 20) if T == 32 else (17,) if T == 64 else (1, 3, 5, 7, 8) if T == 8 else (1, 3, 5, 9, 13, 16)):
        per = 128 * w + 128 * T
        n = int(args.gb * 1e9 / per)
        un = torch.empty(n * 128 * T, dtype=torch.uint8, device=dev)
        pk = torch.empty(n * 128 * w, dtype=torch.uint8, device=dev)
        assert lib.fl_fill_random(un.data_ptr(), un.numel() & ~7, 1, None) == 0 and lib.fl_fill_random(pk.data_ptr(), pk.numel() & ~7, 2, None) == 0
        unv, pkv = un.view(TDT[ty]), pk.view(TDT[ty])
        ops = (("unpack", lambda: fl.BitPacking.unpack(w, pkv, output=unv)), ("pack", lambda: fl.BitPacking.pack(w, unv, output=pkv)))
        if args.ffor:
            refs = torch.empty(n * (T // 8), dtype=torch.uint8, device=dev)
            assert lib.fl_fill_random(refs.data_ptr(), refs.numel() & ~7, 3, None) == 0
            refv = refs.view(TDT[ty])
            ops = (("unfor", lambda: fl.FoR.unfor_pack(w, pkv, refv, output=unv)), ("for", lambda: fl.FoR.for_pack(w, unv, refv, output=pkv)))
        for op, f in ops:
            ms = {k: [] for k in POL}
            for k, p in POL.items():
                lib.fl_internal_set_kernel_policy(p); f()
            torch.cuda.synchronize()
            for _ in range(args.reps):
                for k, p in POL.items():
                    lib.fl_internal_set_kernel_policy(p)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); f(); b.record(); b.synchronize()
                    ms[k].append(a.elapsed_time(b))
            lib.fl_internal_set_kernel_policy(0)
            print(f"{op:6s} {ty:3s} W={w:<2d}       " + " ".join(f"{n * per / sorted(v)[len(v) // 2] / 8e9:12.3f}" for v in ms.values()), flush=True)
        del un, pk, unv, pkv
        torch.cuda.empty_cache()
