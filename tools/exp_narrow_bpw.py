import sys, os, statistics, torch, numpy as np
sys.path.insert(0, os.getcwd())
import fastlanes_amd as fl
from fastlanes_amd import placement as pl
lib = fl.load(); dev = torch.device("cuda:0")
TDT = {"u8": torch.uint8, "u16": torch.uint16}
for ty, T in (("u16", 16), ("u8", 8)):
    n = int(12e9 / (128 * T * 1.5))
    g = torch.Generator(device=dev); g.manual_seed(31 + T)
    widths = torch.randint(1, T, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.uint8)
    offsets, total = fl.widths_to_offsets(ty, widths); pb = int(total)
    for layout, op in (("interleaved", "unpack"), ("interleaved", "unfor"), ("interleaved", "pack"), ("separate", "pack")):
        pack = op == "pack"
        pair = pl.ColumnPair(n * 128 * T, pb, dev, layout=layout) if pack else pl.ColumnPair(pb, n * 128 * T, dev, layout=layout)
        assert lib.fl_fill_random(pair.input.data_ptr(), pair.input.numel() & ~7, 5, None) == 0
        col, un = (pair.output.view(TDT[ty]), pair.input.view(TDT[ty])) if pack else (pair.input.view(TDT[ty]), pair.output.view(TDT[ty]))
        refs = torch.ones(n, dtype=TDT[ty], device=dev)
        pols = [("default", 0)] + [(f"w{w} bpw{b}{' pf' if p else ''}", 2 + 256 * w + 65536 * b + (1 << 24) * p) for w in (8,) for b, p in ((1, 0), (2, 1), (4, 1), (8, 1))]
        res = {k: [] for k, _ in pols}
        for r in range(5):
            for name, pol in pols:
                lib.fl_internal_set_kernel_policy(pol)
                ms = []
                for i in range(6):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    if op == "unpack": fl.unpack_widths(widths, offsets, col, output=un, check=False)
                    elif op == "unfor": fl.unfor_pack_widths(widths, offsets, col, refs, output=un, check=False)
                    else: fl.pack_widths(widths, offsets, un, col, check=False)
                    b.record(); b.synchronize()
                    if i: ms.append(a.elapsed_time(b))
                res[name].append((pb + n * 128 * T) / statistics.median(ms) / 8e9)
        lib.fl_internal_set_kernel_policy(0)
        print(f"{ty} {op}_widths n={n} {layout} {pair.classes}")
        for name, _ in pols: print(f"   {name:14s} {statistics.median(res[name]):.3f}")
        pair.free()
