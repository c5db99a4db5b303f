import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
import fastlanes_amd as fl
from fastlanes_amd import placement as pl
lib = fl.load(); dev = torch.device("cuda:0")
TD = {"u32": (torch.uint32, 32), "u64": (torch.uint64, 64)}
for ty, W in (("u64", 33), ("u32", 20), ("u32", 12)):
    tdt, T = TD[ty]
    n = min(10_000_000, int(40e9 / (128 * W + 128 * T + 128)))
    extra = 2 << 30
    pair = pl.ColumnPair(n * 128 * W, n * 128 * T + extra, dev, aux_bytes=n * 128, layout="interleaved")
    lib.fl_fill_random(pair.input.data_ptr(), (n * 128 * W) & ~7, 3, None); lib.fl_fill_random(pair.aux.data_ptr(), n * 128, 4, None)
    pk, out = pair.input.view(tdt), pair.output[:n * 128 * T].view(tdt)
    sep = torch.empty(n * 128, dtype=torch.uint8, device=dev); sep.copy_(pair.aux)
    tail = pair.output[n * 128 * T + (1 << 30):][:n * 128]; tail.copy_(pair.aux)
    variants = {"unpack": None, "bases with the input (same class)": pair.aux.view(tdt), "bases in their own hipMalloc": sep.view(tdt), "bases behind the output (its classes)": tail.view(tdt)}
    res = {k: [] for k in variants}
    for r in range(5):
        for k, b in variants.items():
            ms = []
            for i in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if b is None: fl.BitPacking.unpack(W, pk, output=out)
                else: fl.Delta.undelta_pack(W, pk, b, output=out)
                e1.record(); e1.synchronize()
                if i: ms.append(e0.elapsed_time(e1))
            nb = n * (128 * W + 128 * T + (0 if b is None else 128))
            res[k].append(nb / statistics.median(ms) / 8e9)
    print(f"{ty} W={W} n={n} classes {pair.classes}")
    for k in variants: print(f"   {k:40s} {statistics.median(res[k]):.3f}")
    pair.free()
