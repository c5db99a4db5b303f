#!/usr/bin/env python3
"""How much of a write-dominated pair's output may come from the INPUT's class of memory?  fl_column_pair_alloc's arrangement search charges a fee
per chunk of the output taken from outside the rotation (fl_capi.hip: choose_chunks; 300 is about break-even with the balance one such chunk buys, 600 is shipped);
FL_INTERNAL_OWN_CLASS_FEE sets it.  One process per fee (the fee is read once), same box, headline shape; prints the class map and the rate.
    for f in 300 1000 150; do FL_INTERNAL_OWN_CLASS_FEE=$f python tools/exp_own_class_fee.py; done"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from fastlanes_amd import placement as pl
lib = fl.load(); dev = torch.device("cuda:0")
for name, T, W, n, tdt in (("unpack u32 W=7", 32, 7, 10_000_000, torch.uint32), ("unpack u16 W=3", 16, 3, 10_000_000, torch.uint16), ("unpack u64 W=9", 64, 9, 5_000_000, torch.uint64)):
    ib, ob = n * 128 * W, n * 128 * T
    res = []
    for rep in range(3):
        pair = pl.ColumnPair(ib, ob, dev, layout="interleaved")
        assert lib.fl_fill_random(pair.input.data_ptr(), ib & ~7, 5, None) == 0
        ms = []
        for i in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fl.BitPacking.unpack(W, pair.input.view(tdt), output=pair.output.view(tdt)); b.record(); b.synchronize()
            if i: ms.append(a.elapsed_time(b))
        res.append(((ib + ob) / statistics.median(ms) / 8e9, pair.classes))
        pair.free()
    print(f"fee {os.environ.get('FL_INTERNAL_OWN_CLASS_FEE', '600 (shipped)'):14s} {name:15s} " + "  ".join(f"{r:.4f}" for r, _ in res) + f"   {res[-1][1][:60]}", flush=True)
