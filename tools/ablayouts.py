#!/usr/bin/env python3
"""The three layouts of fl_column_pair_alloc (include/fastlanes_amd.h) against each other: one process, all three pairs alive at once,
the codec kernel launched on them round-robin (VERDICT r05 "next" #1: the constructed layout must be >= the better of {separate, zoned}
on every box).
    python tools/ablayouts.py [--rounds 5] [--launches 8] [--cases headline,config5,config4,config3]
Prints, per case, the median fraction of the 8 TB/s per layout and the measured class map of the constructed pair."""
import argparse
import os
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402
from fastlanes_amd import placement as pl  # noqa: E402

dev = torch.device("cuda:0")
lib = fl.load()
LAYOUTS = ("interleaved", "separate", "zoned")        # allocation order: the constructed pair's transient pool first


def fill(t, seed):
    if t.numel() & ~7:
        assert lib.fl_fill_random(t.data_ptr(), t.numel() & ~7, seed, None) == 0


def case(name):
    """(label, in_bytes, aux_bytes, out_bytes, make_step(pair) -> callable)"""
    if name == "headline":
        n, w = 10_000_000, 7
        mk = lambda p: (lambda: fl.BitPacking.unpack(w, p.input.view(torch.uint32), output=p.output.view(torch.uint32)))
        return f"unpack u32 W={w}, {n} blocks (BASELINE configs[1])", n * 128 * w, 0, n * 4096, mk
    if name == "config4":
        n, w = 10_000_000, 12
        mk = lambda p: (lambda: fl.Delta.undelta_pack(w, p.input.view(torch.uint32), p.aux.view(torch.uint32), output=p.output.view(torch.uint32)))
        return f"undelta_pack u32 W={w}, {n} blocks (configs[3])", n * 128 * w, n * 128, n * 4096, mk
    if name == "config3":
        n, w = 10_000_000, 17
        mk = lambda p: (lambda: fl.BitPacking.unpack(w, p.input.view(torch.uint64), output=p.output.view(torch.uint64)))
        return f"unpack u64 W={w}, {n} blocks (configs[2], decode leg)", n * 128 * w, 0, n * 8192, mk
    if name == "config5":
        n = 9_765_625
        widths = torch.from_numpy((1 + np.arange(n) % 32).astype(np.uint8)).to(dev)
        offsets, total = fl.widths_to_offsets("u32", widths)

        def mk(p):
            return lambda: fl.unpack_widths(widths, offsets, p.input.view(torch.uint32), output=p.output.view(torch.uint32), check=False)
        return f"unpack u32 width[b] = 1 + b mod 32, {n} blocks (configs[4])", int(total), 0, n * 4096, mk
    raise SystemExit(f"unknown case {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--launches", type=int, default=8)
    ap.add_argument("--cases", default="headline,config5")
    a = ap.parse_args()
    print(f"# {torch.cuda.get_device_name(0)}; every layout's pair alive at once, {a.rounds} rounds x {a.launches} launches round-robin; "
          "fraction of 8 TB/s (SURVEY.md 8d bytes), median over rounds of each round's median launch")
    for name in a.cases.split(","):
        label, ib, ab, ob, mk = case(name)
        pairs, steps = {}, {}
        for lay in LAYOUTS:
            try:
                pairs[lay] = pl.ColumnPair(ib, ob, dev, aux_bytes=ab, layout=lay)
            except Exception as e:                              # e.g. the zoned slab does not fit next to the others
                print(f"  ({lay}: not allocated: {e})")
                continue
            fill(pairs[lay].input, 11)
            fill(pairs[lay].aux, 12)
            steps[lay] = mk(pairs[lay])
        rates = {lay: [] for lay in steps}
        for r in range(a.rounds):
            for lay, step in steps.items():
                ms = []
                for i in range(a.launches + 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    step()
                    e1.record()
                    e1.synchronize()
                    if i:
                        ms.append(e0.elapsed_time(e1))
                rates[lay].append((ib + ab + ob) / statistics.median(ms) / 1e6 / 8000)
        print(f"{label}")
        for lay in steps:
            print(f"  {lay:12s} {statistics.median(rates[lay]):.3f}   per round: " + " ".join(f"{x:.3f}" for x in rates[lay])
                  + (f"   classes {pairs[lay].classes}" if lay == "interleaved" else ""))
        best_other = max((statistics.median(rates[l]) for l in steps if l != "interleaved"), default=0.0)
        if "interleaved" in steps:
            print(f"  constructed / better of the others = {statistics.median(rates['interleaved']) / best_other:.3f}" if best_other else "")
        for p in pairs.values():
            p.free()
        pairs.clear()
        steps.clear()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
