#!/usr/bin/env python3
"""A/B of the tile-map window (fl_kernels.hpp: xcd_tile) for every kernel family, same buffers, launches interleaved round-robin.
    python tools/abwindow.py [--reps 9] [--gb 48] [--windows 31,12,14,16,18,20] [--cases matrix] [--ops pack,delta] [--constructed]
Window = log2 of the window in 1024-value blocks (fastlanes_amd_internal.h: policy bits 25-29); 31 = one window = the whole-column
map of rounds 1-3; 0 = the library's own choice per kernel.  Prints GB/s of algorithmic bytes per (op, window)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlanes_amd as fl  # noqa: E402

ESZ = {"u8": 1, "u16": 2, "u32": 4, "u64": 8}
TDT = {"u8": torch.uint8, "u16": torch.uint16, "u32": torch.uint32, "u64": torch.uint64}
dev = torch.device("cuda:0")
lib = fl.load()


def filled(nbytes, seed):
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if nbytes & ~7:
        assert lib.fl_fill_random(t.data_ptr(), nbytes & ~7, seed, None) == 0
    return t


CONSTRUCTED = False
_pairs = []


def buffers(in_bytes, out_bytes, aux_bytes=0):
    """(input filled with random bits, aux likewise, output) -- plain tensors, or with --constructed one fl_column_pair_alloc(INTERLEAVED)
    pair per case (the input + aux inside one class of memory, the output arranged for the eight XCDs' write positions: DESIGN.md section 4)"""
    if not CONSTRUCTED:
        return filled(in_bytes, 1), (filled(aux_bytes, 2) if aux_bytes else None), torch.empty(out_bytes, dtype=torch.uint8, device=dev)
    from fastlanes_amd import placement as pl
    while _pairs:
        _pairs.pop().free()
    pair = pl.ColumnPair(in_bytes, out_bytes, dev, aux_bytes=aux_bytes, layout="interleaved")
    _pairs.append(pair)
    assert lib.fl_fill_random(pair.input.data_ptr(), in_bytes & ~7, 1, None) == 0
    if aux_bytes:
        assert lib.fl_fill_random(pair.aux.data_ptr(), aux_bytes & ~7, 2, None) == 0
    return pair.input, (pair.aux if aux_bytes else None), pair.output


def case(op, ty, w, gb):
    """(label, algorithmic bytes per launch, callable)"""
    T, esz = ESZ[ty] * 8, ESZ[ty]
    un, pk = 128 * T, 128 * w
    per = {"unpack": pk + un, "pack": pk + un, "undelta_pack": pk + 128 + un, "delta": 2 * un + 128, "undelta": 2 * un + 128,
           "transpose": 2 * un, "untranspose": 2 * un, "unpack_compare": pk + 128, "unpack_block_sums": pk + 8, "block_min_max": un + 2 * esz,
           "unpack_mixed": 128 * (T + 1) / 2 + un, "transpose_delta_pack": pk + 128 + un, "undelta_pack_untranspose": pk + 128 + un}[op]
    n = min(10_000_000, int(gb * 1e9 / per))
    v = lambda t: t.view(TDT[ty])
    if op in ("unpack", "undelta_pack", "undelta_pack_untranspose"):
        src, bases, dst = buffers(n * pk, n * un, 0 if op == "unpack" else n * 128)
        src, dst = v(src), v(dst)
        if op == "unpack":
            f = lambda: fl.BitPacking.unpack(w, src, output=dst)
        else:
            bases = v(bases)
            g = getattr(fl.Delta, op)
            f = lambda: g(w, src, bases, output=dst)
    elif op in ("pack", "transpose_delta_pack"):
        src, bases, dst = buffers(n * un, n * pk, 0 if op == "pack" else n * 128)
        src, dst = v(src), v(dst)
        if op == "pack":
            f = lambda: fl.BitPacking.pack(w, src, output=dst)
        else:
            bases = v(bases)
            f = lambda: fl.Delta.transpose_delta_pack(w, src, bases, output=dst)
    elif op in ("delta", "undelta"):
        src, bases, dst = (v(t) for t in buffers(n * un, n * un, n * 128))
        g = getattr(fl.Delta, op)
        f = lambda: g(src, bases, output=dst)
    elif op in ("transpose", "untranspose"):
        src, _, dst = buffers(n * un, n * un)
        src, dst = v(src), v(dst)
        g = getattr(fl.Transpose, op)
        f = lambda: g(src, output=dst)
    elif op == "unpack_compare":
        src, mask = v(filled(n * pk, 1)), torch.empty(n * 32, dtype=torch.int32, device=dev)
        f = lambda: fl.BitPacking.unpack_compare(w, src, "<", (1 << w) // 2, output=mask)
    elif op == "unpack_block_sums":
        src, sums = v(filled(n * pk, 1)), torch.empty(n, dtype=torch.int64, device=dev)
        f = lambda: fl.BitPacking.unpack_block_sums(w, src, output=sums)
    elif op == "block_min_max":
        src = v(filled(n * un, 1))
        mm = (v(torch.empty(n * esz, dtype=torch.uint8, device=dev)), v(torch.empty(n * esz, dtype=torch.uint8, device=dev)))
        f = lambda: fl.BitPacking.block_min_max(src, output=mm)
    else:  # unpack_mixed: width[b] = 1 + b mod T
        widths = (1 + torch.arange(n, dtype=torch.int64, device=dev) % T).to(torch.uint8)
        offsets, total = fl.widths_to_offsets(ty, widths)
        src, dst = v(filled(int(total.item()), 1)), v(torch.empty(n * un, dtype=torch.uint8, device=dev))
        per = (int(total.item()) + n * un) / n
        f = lambda: fl.unpack_widths(widths, offsets, src, output=dst, check=False)
    return f"{op} {ty} W={w if op != 'unpack_mixed' else '1..T'} ({n} blocks)", n * per, f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--gb", type=float, default=48.0)
    ap.add_argument("--windows", default="31,12,14,16,18,20")
    ap.add_argument("--cases", default="all")
    ap.add_argument("--ops", default="", help="keep only these ops of the chosen cases")
    ap.add_argument("--constructed", action="store_true",
                    help="every materialising case's buffers from fl_column_pair_alloc(FL_LAYOUT_INTERLEAVED) instead of plain tensors")
    args = ap.parse_args()
    global CONSTRUCTED
    CONSTRUCTED = args.constructed
    windows = [int(x) for x in args.windows.split(",")]
    cases = [("unpack", "u32", 7), ("pack", "u32", 7), ("unpack", "u64", 17), ("pack", "u64", 17), ("undelta_pack", "u32", 12),
             ("unpack_mixed", "u32", 0), ("unpack", "u16", 3), ("pack", "u16", 3), ("unpack", "u8", 3), ("pack", "u8", 3),
             ("delta", "u32", 0), ("undelta", "u32", 0), ("transpose", "u32", 0), ("untranspose", "u64", 0),
             ("transpose_delta_pack", "u32", 12), ("transpose_delta_pack", "u64", 20), ("transpose_delta_pack", "u16", 9),
             ("transpose_delta_pack", "u8", 4), ("undelta_pack_untranspose", "u32", 12),
             ("unpack_compare", "u16", 3), ("unpack_compare", "u32", 7), ("unpack_compare", "u64", 17),
             ("unpack_block_sums", "u32", 7), ("unpack_block_sums", "u16", 3), ("block_min_max", "u32", 0)]
    if args.cases == "matrix":
        # every row of fl_window_table.inc x every element type (what tools/make_window_table.py reads), one mid-range width each
        W1 = {"u8": 3, "u16": 3, "u32": 7, "u64": 17}          # BitPacking
        W2 = {"u8": 4, "u16": 9, "u32": 12, "u64": 20}         # Delta
        cases = []
        for ty in ("u8", "u16", "u32", "u64"):
            cases += [(op, ty, W1[ty]) for op in ("unpack", "pack", "unpack_compare", "unpack_block_sums")]
            cases += [(op, ty, W2[ty]) for op in ("undelta_pack", "undelta_pack_untranspose", "transpose_delta_pack")]
            cases += [(op, ty, 0) for op in ("delta", "undelta", "transpose", "untranspose", "block_min_max")]
    elif args.cases != "all":
        keep = args.cases.split(",")
        cases = [c for c in cases if c[0] in keep]
    if args.ops:
        cases = [c for c in cases if c[0] in args.ops.split(",")]
    print(f"# {lib.fl_version().decode()}" + ("  [buffers: constructed pairs]" if CONSTRUCTED else "") + f"\n# GB/s of algorithmic bytes (fraction of 8 TB/s), median of {args.reps} round-robin launches per window; window = log2 blocks, "
          "31 = whole column (rounds 1-3), 0 = the library's per-kernel default", flush=True)
    print(f"{'case':58s} " + " ".join(f"{('w=' + str(x)):>13s}" for x in [0] + windows), flush=True)
    for op, ty, w in cases:
        label, nbytes, f = case(op, ty, w, args.gb)
        ms = {x: [] for x in [0] + windows}
        for x in [0] + windows:           # warm every variant
            lib.fl_internal_set_kernel_policy(x << 25)
            f()
        torch.cuda.synchronize()
        for _ in range(args.reps):
            for x in [0] + windows:
                lib.fl_internal_set_kernel_policy(x << 25)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); b.synchronize()
                ms[x].append(a.elapsed_time(b))
        lib.fl_internal_set_kernel_policy(0)
        row = []
        for x in [0] + windows:
            t = sorted(ms[x])[len(ms[x]) // 2]
            row.append(f"{nbytes / t / 1e6:6.0f} ({nbytes / t / 8e9:.3f})")
        print(f"{label:58s} " + " ".join(row), flush=True)
        del f
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
