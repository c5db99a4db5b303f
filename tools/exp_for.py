"""Scratch (GPU box): for_pack / unfor_pack through both kernel designs on the same buffers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlanes_amd as fl
from bench import rand_u8
lib = fl.load()
dev = torch.device("cuda", 0)
n = 4_000_000
for ty, tdt, T, W in (("u32", torch.uint32, 32, 7), ("u32", torch.uint32, 32, 16), ("u64", torch.uint64, 64, 17), ("u16", torch.uint16, 16, 9)):
    esz = T // 8
    nn = n * 4 // esz
    un = rand_u8(nn * 1024 * esz, 1, dev).view(tdt)
    pk = rand_u8(nn * 128 * W, 2, dev).view(tdt)
    refs = rand_u8(nn * esz, 3, dev).view(tdt)
    one = refs[:1]
    bytes_ = nn * (128 * W + 1024 * esz)
    cases = {
        "pack": lambda: fl.BitPacking.pack(W, un, output=pk),
        "for_pack refs[b]": lambda: fl.FoR.for_pack(W, un, refs, output=pk),
        "for_pack one ref": lambda: fl.FoR.for_pack(W, un, one, output=pk),
        "unpack": lambda: fl.BitPacking.unpack(W, pk, output=un),
        "unfor_pack refs[b]": lambda: fl.FoR.unfor_pack(W, pk, refs, output=un),
        "unfor_pack one ref": lambda: fl.FoR.unfor_pack(W, pk, one, output=un),
    }
    res = {}
    for rnd in range(4):
        for pol in (1, 2):
            lib.fl_set_kernel_policy(pol)
            for name, f in cases.items():
                f(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); torch.cuda.synchronize()
                res.setdefault((name, pol), []).append(a.elapsed_time(b))
    for name in cases:
        c, w = sorted(res[(name, 1)])[1], sorted(res[(name, 2)])[1]
        print(f"{ty} W={W:2d} {name:20s} cell-column {bytes_ / c / 1e6:7.1f} GB/s   wave-per-block {bytes_ / w / 1e6:7.1f} GB/s   auto={lib.fl_get_kernel_policy()}", flush=True)
    lib.fl_set_kernel_policy(0)
    del un, pk, refs
