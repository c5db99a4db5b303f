#!/usr/bin/env python3
"""Does it matter where a mixed-width column's widths[] / offsets[] live?  (9 bytes per block: a thin READ stream, but the block's packed rows
cannot be requested before it has arrived.)  The same unpack_widths call with the two arrays (a) in plain tensors, (b) in the constructed
pair's aux buffer = the input's class of memory, (c) at the end of the pair's OUTPUT range; launches round-robin.
    python tools/exp_meta_placement.py [u32,u64,u16,u8]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from fastlanes_amd import placement as pl

lib = fl.load(); dev = torch.device("cuda:0")
TDT = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
BITS = {"u8": 8, "u16": 16, "u32": 32, "u64": 64}
for ty in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["u32", "u64", "u16", "u8"]):
    T = BITS[ty]
    n = int(60e9 / (128 * T * 1.5)) if ty == "u32" else int(24e9 / (128 * T * 1.5))
    if ty == "u32":
        n = 9_765_625
        widths = (1 + torch.arange(n, dtype=torch.int64, device=dev) % 32).to(torch.uint8)          # BASELINE config 5
    else:
        g = torch.Generator(device=dev); g.manual_seed(31 + T)
        widths = torch.randint(1, T, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.uint8)
    offsets, total = fl.widths_to_offsets(ty, widths); pb = int(total)
    meta = (n + 255) & ~255
    slack = 1 << 20
    pair = pl.ColumnPair(pb, n * 128 * T + meta + 8 * n + slack, dev, aux_bytes=meta + 8 * n, layout="interleaved")
    assert lib.fl_fill_random(pair.input.data_ptr(), pb & ~7, 5, None) == 0
    col, un = pair.input.view(TDT[ty]), pair.output[:n * 128 * T].view(TDT[ty])
    w_aux, o_aux = pair.aux[:n], pair.aux[meta:meta + 8 * n].view(torch.int64)
    w_aux.copy_(widths); o_aux.copy_(offsets)
    tail = pair.output[n * 128 * T + slack // 2:]
    w_out, o_out = tail[:n], tail[meta:meta + 8 * n].view(torch.int64)
    w_out.copy_(widths); o_out.copy_(offsets)
    variants = [("plain tensors", widths, offsets), ("the pair's aux (input's class)", w_aux, o_aux), ("behind the pair's output", w_out, o_out)]
    ms = [[] for _ in variants]
    for r in range(8):
        for k, (_, w, o) in enumerate(variants):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fl.unpack_widths(w, o, col, output=un, check=False); b.record(); b.synchronize()
            if r: ms[k].append(a.elapsed_time(b))
    print(f"{ty} n={n} {pair.classes[:60]}")
    for (name, _, _), m in zip(variants, ms):
        print(f"   widths[] / offsets[] in {name:32s} {(pb + n * 128 * T) / statistics.median(m) / 8e9:.4f}", flush=True)
    pair.free()
